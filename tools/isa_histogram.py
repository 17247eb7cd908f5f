#!/usr/bin/env python3
"""Static VALU instruction mix of the raster / binning kernels -> profiles/<round>_isa_histogram.json.

bench.py prices a kernel's VALU issue time as  sum over classes( SQ_INSTS_VALU x share of the class x measured cycles per wave-instruction of
the class )  — the shares come from here (the static mix of the kernel's ISA: a proxy for the dynamic one), the rates from
tools/valu_rate.hip (profiles/<round>_valu_rate.json). Usage: tools/isa_histogram.py <round>   (needs hipcc; no GPU)"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402
from contrast_renderer_amd.build import FLAGS  # noqa: E402

KERNELS = {"raster_edges.hip": ["k_raster_fill", "k_raster_edges", "k_raster_rows", "k_bin_flat", "k_bin_edges", "k_scatter"], "raster.hip": ["k_raster_tile", "k_prim_setup"], "tessellate.hip": ["k_count", "k_emit", "k_hull_small"]}


def classify(m):
    if not m.startswith("v_"):
        return None
    if m.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    if m.startswith("v_pk_"):
        return "packed_f32"
    if m.startswith("v_cmp") or m.startswith("v_cmpx"):
        return "compare"
    if m.startswith("v_cndmask"):
        return "cndmask"
    if "f64" in m:
        return "f64"
    if m.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "transcendental"
    if m.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64")):
        return "quarter_rate_int"
    if m.startswith("v_cvt"):
        return "convert"
    return "simple"


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
    out = {"kernel_source_hash": kernel_source_hash(), "note": "static counts of VALU mnemonics per kernel (hipcc -S of the shipped sources with the shipped flags)", "kernels": {}}
    for src, names in KERNELS.items():
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            flags = [f for f in FLAGS if f != "-fPIC"]
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", os.path.join(ROOT, "contrast_renderer_amd", "csrc", src), "-o", tmp.name], stderr=subprocess.DEVNULL)
            text = open(tmp.name).read()
        for block in re.split(r"\n(?=_Z[\w]+:)", text):
            head = block.split(":", 1)[0]
            if not head.startswith("_Z") or not any(n in head for n in names):
                continue
            demangled = subprocess.run(["c++filt", head], capture_output=True, text=True).stdout.strip().split("(")[0]
            body = block.split(".Lfunc_end")[0]
            hist = collections.Counter()
            salu = 0
            for line in body.splitlines():
                m = re.match(r"\s+([a-z_0-9]+)\b", line)
                if not m:
                    continue
                c = classify(m.group(1))
                if c:
                    hist[c] += 1
                elif m.group(1).startswith("s_") and not m.group(1).startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm")):
                    salu += 1
            out["kernels"][demangled.replace("void ", "")] = {"valu": dict(hist), "valu_total": sum(hist.values()), "salu_total": salu}
    path = os.path.join(ROOT, "profiles", f"{rnd}_isa_histogram.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)
    for k, v in out["kernels"].items():
        print(f"{k:50s}", v["valu_total"], v["valu"])


if __name__ == "__main__":
    main()
