"""GPU box: host time of every call of a consumed animation step (two targets), averaged: where does the host block?"""
import sys, time, os
import numpy as np
os.environ.setdefault("CRH_EDGE_PASS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene
sc = scenes.scene_cubic_fill(10000)
r = Renderer(Configuration(1, 4, 4), device=0)
scene = Scene(r, sc["batch"]); shown = [Frame(r, 4096, 4096), Frame(r, 4096, 4096)]
tr, co = sc["transforms"], sc["colors"]
scene.set_instances(tr, co)
moved = []
for k in range(20):
    t = np.array(tr, dtype=np.float32, copy=True).reshape(-1, 16); t[:, [0, 1, 4, 5, 12, 13]] *= np.float32(1.01 ** (k if k <= 10 else 20 - k)); moved.append(t)
def loop(n, mode, acc):
    for i in range(n):
        f = shown[i % 2]
        t0 = time.perf_counter(); f.synchronize()
        t1 = time.perf_counter(); scene.tessellate()  # (first: it does not depend on the instances, and started early it runs in the gap behind the raster kernel of the frame before)
        t2 = time.perf_counter()
        if mode == "same": scene.set_instances(tr, co)
        if mode == "moved": scene.set_instances(moved[i % 20], co)
        t3 = time.perf_counter(); f.clear()
        t4 = time.perf_counter(); scene.render(f)
        t5 = time.perf_counter()
        for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[k] += d
for mode in ("steady", "same", "moved"):
    if mode == "steady": scene.set_instances(tr, co)
    loop(80, mode, [0] * 5); r.synchronize()
    acc = [0.0] * 5; t0 = time.perf_counter(); loop(60, mode, acc); r.synchronize(); total = (time.perf_counter() - t0) / 60
    print(mode, "%.4f ms/step;" % (total * 1e3), "host ms/step: synchronize %.3f tessellate %.3f set_instances %.3f clear %.3f render %.3f" % tuple(a / 60 * 1e3 for a in acc), flush=True)
