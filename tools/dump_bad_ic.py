import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from golden_util import load_golden
from contrast_renderer_amd import renderer as R
from oracle import Oracle
from oracle.binding import split_shape
batch, z = load_golden("cubic_fill_40")
r = R.Renderer(R.Configuration(), 0)
scene = R.Scene(r, batch); o = Oracle(batch)
cd = batch.control_data.reshape(40, -1); lines = []
for s in range(40):
    g = split_shape(*scene.shape(s)); w = split_shape(*o.shape(s))
    if not np.array_equal(g["integral_cubic"], w["integral_cubic"]):
        a = g["integral_cubic"].view(np.float32).reshape(-1, 5); b = w["integral_cubic"].view(np.float32).reshape(-1, 5)
        rows = sorted(set(np.nonzero(a != b)[0]))
        print("shape", s, "rows", rows)
        for rr in rows[:3]: print("   gpu", a[rr], "oracle", b[rr])
        prev = batch.path_start[s]; off = 0
        for k in range(8):
            t = int(batch.segment_types[s * 8 + k]); n = [2, 4, 6, 5, 10][t]; rec = cd[s, off:off + n]; off += n
            if t == 2: lines.append(" ".join(repr(float(x)) for x in list(prev) + list(rec)))
            prev = rec[-2:]
        if len(lines) >= 12: break
open("gpurun_out/bad_ic.txt", "w").write("\n".join(lines))
