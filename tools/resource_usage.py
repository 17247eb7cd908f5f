#!/usr/bin/env python3
"""Host: compile one csrc/*.hip with -Rpass-analysis=kernel-resource-usage and print one line per kernel
(VGPRs, AGPRs, SGPRs, scratch bytes per lane, occupancy, LDS). Usage: tools/resource_usage.py raster_edges.hip [extra hipcc flags...]"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from contrast_renderer_amd.build import CSRC, FLAGS, file_flags  # noqa: E402


def usage(src, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + file_flags().get(src, []) + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, "x.o")]
        err = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: +(\w[\w ]*?): +(\S+)", line) or re.search(r"remark: +(\w[\w ]*?): +(\S+)", line)
        if "Function Name" in line:
            cur = {"name": line.split("Function Name:")[1].split("[")[0].strip()}
            rows.append(cur)
        elif cur is not None:
            m = re.search(r"(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): +(\d+)", line)
            if m:
                cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return rows


if __name__ == "__main__":
    src = sys.argv[1]
    for row in usage(src, sys.argv[2:]):
        name = subprocess.run(["/usr/bin/c++filt", row["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        print(f"{name:58s} vgpr {row.get('VGPRs', 0):4d} agpr {row.get('AGPRs', 0):3d} sgpr {row.get('TotalSGPRs', 0):4d} scratch {row.get('ScratchSize', 0):4d} occ {row.get('Occupancy', 0):2d} lds {row.get('LDS', 0):6d}")
