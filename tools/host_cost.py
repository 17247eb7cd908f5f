import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
from contrast_renderer_amd import scenes, _ffi
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene, RenderOperation as Op
sc = scenes.scene_cubic_fill(10000)
r = Renderer(Configuration(1, 4, 4), device=0)
scene = Scene(r, sc["batch"]); frame = Frame(r, 4096, 4096)
draws = np.array([d for i in range(10000) for d in ((i, i, int(Op.Stencil), 0, 0), (i, i, int(Op.Color), 0, 0))], dtype=np.uint32)
t = np.ascontiguousarray(sc["transforms"], dtype=np.float32); c = np.ascontiguousarray(sc["colors"], dtype=np.float32)
fp = C.POINTER(C.c_float)
for it in range(3):
    frame.clear(); scene.render_draws(frame, t, c, draws)
r.synchronize()
ts = []
for it in range(30):
    frame.clear()
    t0 = time.perf_counter()
    rc = r.lib.crh_scene_render_draws(scene.handle, frame.handle, t.ctypes.data_as(fp), c.ctypes.data_as(fp), len(t), draws.ctypes.data_as(C.POINTER(_ffi.DrawC)), len(draws))
    ts.append(time.perf_counter() - t0)
r.synchronize()
print("C call host time: median %.3f ms, min %.3f" % (np.median(ts) * 1e3, min(ts) * 1e3))
scene.set_instances(t, c)
ts = []
for it in range(30):
    frame.clear()
    t0 = time.perf_counter(); r.lib.crh_scene_set_instances(scene.handle, t.ctypes.data_as(fp), c.ctypes.data_as(fp)); t1 = time.perf_counter()
    r.lib.crh_scene_render_resident(scene.handle, frame.handle); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1))
r.synchronize()
print("set_instances %.3f ms, render_resident %.3f ms" % tuple(np.median(np.array(ts), axis=0) * 1e3))
ts = []
for it in range(30):
    r.synchronize()
    t0 = time.perf_counter(); r.lib.crh_scene_set_instances(scene.handle, t.ctypes.data_as(fp), c.ctypes.data_as(fp)); t1 = time.perf_counter()
    ts.append(t1 - t0)
print("set_instances with the GPU idle %.3f ms" % (np.median(ts) * 1e3))
small = scenes.scene_cubic_fill(100)
s2 = Scene(r, small["batch"]); t2_ = np.ascontiguousarray(small["transforms"]); c2_ = np.ascontiguousarray(small["colors"])
ts = []
for it in range(30):
    r.synchronize()
    t0 = time.perf_counter(); r.lib.crh_scene_set_instances(s2.handle, t2_.ctypes.data_as(fp), c2_.ctypes.data_as(fp)); t1 = time.perf_counter()
    ts.append(t1 - t0)
print("set_instances of 100 shapes, GPU idle %.3f ms" % (np.median(ts) * 1e3))
