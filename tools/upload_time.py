import time, numpy as np, sys
sys.path.insert(0, '/root/repo')
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene
sc = scenes.scene_cubic_fill(10000)
r = Renderer(Configuration(1, 4, 4), device=0)
scene = None
for i in range(5):
    t0 = time.perf_counter()
    scene = Scene(r, sc["batch"], tessellate=False, existing=scene)
    t1 = time.perf_counter()
    scene.tessellate(); r.synchronize()
    t2 = time.perf_counter()
    print("upload %.3f ms  tessellate+sync %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
