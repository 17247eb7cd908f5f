"""GPU box: where does a consumed-frame loop lose its time? one target / two targets, with and without crh_frame_synchronize before reuse."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene
sc = scenes.scene_cubic_fill(10000)
r = Renderer(Configuration(1, 4, 4), device=0)
scene = Scene(r, sc["batch"]); scene.set_instances(sc["transforms"], sc["colors"])
targets = [Frame(r, 4096, 4096) for _ in range(3)]
def loop(n, k, consume):
    host = 0.0
    for i in range(n):
        f = targets[i % k]
        t0 = time.perf_counter()
        if consume: f.synchronize()
        t1 = time.perf_counter()
        scene.tessellate(); f.clear(); scene.render(f)
        host += time.perf_counter() - t1
    return host / n
for k, consume in ((1, False), (2, False), (2, True), (3, True), (1, True), (2, True)):
    loop(60, k, consume); r.synchronize()
    t0 = time.perf_counter(); h = loop(60, k, consume); r.synchronize()
    print(f"targets {k} consume {consume}: {(time.perf_counter() - t0) / 60 * 1e3:.4f} ms/step, host time in the submit calls {h * 1e3:.4f} ms/step", flush=True)
