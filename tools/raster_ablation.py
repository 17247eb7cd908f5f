import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from contrast_renderer_amd import scenes, renderer as R
sc = scenes.scene_cubic_fill(10000)
r = R.Renderer(R.Configuration(), 0)
scene = R.Scene(r, sc["batch"]); scene.check(); scene.set_instances(sc["transforms"], sc["colors"])
frame = R.Frame(r, 4096, 4096)
frame.clear(); scene.render(frame); r.synchronize()
out = (C.c_uint32 * 8)(); r.lib.crh_debug_frame_counters(frame.handle, out)
print("band pairs %d (%.1f per band, %.1f per tile)" % (out[1], out[1] / (65536 * 4), out[1] / 65536))
