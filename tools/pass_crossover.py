"""DEVELOPMENT TOOL (GPU): edge pass against triangle pass (CRH_TRIANGLE_PASS=1) by Shape size — where is the crossover?"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrast_renderer_amd import renderer as R, scenes


def ms_per_frame(sc, size, triangle):
    if triangle:
        os.environ["CRH_TRIANGLE_PASS"] = "1"
    else:
        os.environ.pop("CRH_TRIANGLE_PASS", None)
    r = R.Renderer(R.Configuration(1, 4, 4), 0)
    scene = R.Scene(r, sc["batch"]); scene.check(); scene.set_instances(sc["transforms"], sc["colors"])
    frames = [R.Frame(r, size, size) for _ in range(2)]
    def run(n):
        for i in range(n):
            f = frames[i % 2]; f.clear(); scene.render(f)
    run(6); r.synchronize()
    t0 = time.perf_counter(); run(30); r.synchronize()
    return (time.perf_counter() - t0) / 30 * 1e3


for lo, hi in ((2, 6), (4, 12), (6, 20), (8, 32), (12, 48), (16, 64)):
    sc = scenes.scene_cubic_fill(20000, (2048, 2048), r_lo=float(lo), r_hi=float(hi))
    print(f"cubic radius {lo}..{hi} px: edge {ms_per_frame(sc, 2048, False):.3f} ms, triangle {ms_per_frame(sc, 2048, True):.3f} ms (render only, no tessellation)")
for sizes in ((12.0,), (24.0,), (48.0,), (96.0,)):
    sc = scenes.scene_glyphs(20000, (2048, 2048), sizes=sizes)
    print(f"glyphs size {sizes[0]} px: edge {ms_per_frame(sc, 2048, False):.3f} ms, triangle {ms_per_frame(sc, 2048, True):.3f} ms")
