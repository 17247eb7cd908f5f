"""DEVELOPMENT TOOL (GPU, library built with -DCRH_ABLATE): entries of the edge pass by class for a workload."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRH_RASTER_DEBUG"] = "256"
from contrast_renderer_amd import renderer as R, scenes
w = sys.argv[1] if len(sys.argv) > 1 else "cubic"
sc = {"cubic": lambda: scenes.scene_cubic_fill(10000), "glyphs": lambda: scenes.scene_glyphs(50000, (2048, 2048)), "dashed": lambda: scenes.scene_dashed_strokes(2000)}[w]()
size = {"cubic": 4096, "glyphs": 2048, "dashed": 4096}[w]
r = R.Renderer(R.Configuration(4 if w == "dashed" else 1, 4, 4), 0)
scene = R.Scene(r, sc["batch"]); scene.check(); scene.set_instances(sc["transforms"], sc["colors"])
frame = R.Frame(r, size, size)
out = (C.c_uint32 * 16)()
for _ in range(3):  # the first frames learn the list capacity; the counters are read and reset after each
    frame.clear(); scene.render(frame); r.synchronize()
    r.lib.crh_debug_frame_counters16(frame.handle, out)
tiles = (size // 16) ** 2
print(w, "pairs", out[1], "per tile %.1f" % (out[1] / tiles), "longest", out[3], "| fill edges", out[8], "hull edges", out[9], "synth", out[10], "triangles", out[11])
