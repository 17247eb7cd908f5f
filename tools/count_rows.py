"""DEVELOPMENT TOOL (GPU, library built with -DCRH_ABLATE): how often each part of k_raster_rows runs on a workload."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRH_RASTER_DEBUG"] = "256"
from contrast_renderer_amd import renderer as R, scenes
w = sys.argv[1] if len(sys.argv) > 1 else "cubic"
sc = {"cubic": lambda: scenes.scene_cubic_fill(10000), "glyphs": lambda: scenes.scene_glyphs(50000, (2048, 2048)), "s100k": lambda: scenes.scene_cubic_fill(100000, (8192, 8192))}[w]()
size = {"cubic": 4096, "glyphs": 2048, "s100k": 8192}[w]
r = R.Renderer(R.Configuration(1, 4, 4), 0)
scene = R.Scene(r, sc["batch"]); scene.check(); scene.set_instances(sc["transforms"], sc["colors"])
frame = R.Frame(r, size, size)
out = (C.c_uint32 * 16)()
for _ in range(3):
    frame.clear(); scene.render(frame); r.synchronize()
    r.lib.crh_debug_frame_counters16(frame.handle, out)
tiles = (size // 16) ** 2
names = {1: "pairs", 3: "longest", 8: "rounds", 9: "edge iterations", 10: "edge-like walked", 11: "triangles walked", 12: "triangle passes", 13: "covers walked", 14: "tiles with a list", 15: "triangle box samples"}
print(w, {names[i]: (out[i], round(out[i] / tiles, 2)) for i in sorted(names)})
