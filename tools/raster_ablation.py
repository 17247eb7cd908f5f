import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from contrast_renderer_amd import scenes, renderer as R
sc = scenes.scene_cubic_fill(10000)
r = R.Renderer(R.Configuration(), 0)
scene = R.Scene(r, sc["batch"]); scene.check(); scene.set_instances(sc["transforms"], sc["colors"])
frame = R.Frame(r, 4096, 4096)
def run(debug, steps=10):
    os.environ["CRH_RASTER_DEBUG"] = str(debug)
    for _ in range(2): frame.clear(); scene.render(frame)
    r.synchronize(); r.enable_timing(True)
    for _ in range(steps): frame.clear(); scene.render(frame)
    kt = r.kernel_times(); r.enable_timing(False)
    ms = [m for n, m, b in kt if n == "raster_tiles"]
    return sum(ms) / len(ms)
print("full               %.3f ms" % run(0))
print("stop after sort    %.3f ms" % run(1))
print("gather, no coverage %.3f ms" % run(2))
os.environ["CRH_RASTER_DEBUG"] = "4"; frame.clear(); scene.render(frame)
out = (C.c_uint32 * 8)(); r.lib.crh_debug_frame_counters(frame.handle, out)
nt = 65536
print("pairs %d (%.1f/tile, max %d)  candidates %.1f/tile (max %d)  survivors %.1f/tile" % (out[1], out[2] / nt, out[5], out[3] / nt, out[6], out[4] / nt))
