"""DEVELOPMENT TOOL (GPU): the edge pass with and without the opaque-cover shortcut (CRH_RASTER_DEBUG bit 15) on many opaque-heavy scenes."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRH_EDGE_PASS"] = "1"
from contrast_renderer_amd import renderer as R, scenes


def render(sc, colors, size, msaa, off):
    os.environ["CRH_RASTER_DEBUG"] = "32768" if off else "0"
    r = R.Renderer(R.Configuration(msaa, 4, 4), 0)
    scene = R.Scene(r, sc["batch"]); scene.check()
    frame = R.Frame(r, size, size)
    frame.clear(); scene.render(frame, sc["transforms"], colors)
    return frame.download()


bad = 0
rng = np.random.RandomState(3)
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    size = [1024, 2048, 768][seed % 3]
    n = [4000, 12000, 2500][seed % 3]
    hi = [48.0, 90.0, 140.0][(seed // 3) % 3]
    sc = scenes.scene_cubic_fill(n, (size, size), r_lo=5.0, r_hi=hi, config_index=100 + seed)
    colors = np.asarray(sc["colors"], np.float32).copy()
    colors[rng.uniform(size=len(colors)) < 0.8, 3] = 1.0
    msaa = 4 if seed % 4 == 3 else 1
    a, b = render(sc, colors, size, msaa, False), render(sc, colors, size, msaa, True)
    d = int((a != b).any(axis=2).sum())
    bad += d
    print(f"seed {seed}: {n} shapes r<={hi} @ {size}^2 msaa {msaa}: {d} pixels differ")
print("TOTAL", bad)
