"""ms per frame when the instance transforms change every frame (crh_scene_render = set_instances + render, the showcase's loop,
main.rs:154-250) against re-rendering resident instances (GPU box)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene
sc = scenes.scene_cubic_fill(10000)
r = Renderer(Configuration(1, 4, 4), device=0)
scene = Scene(r, sc["batch"])
frame = Frame(r, 4096, 4096)
scene.set_instances(sc["transforms"], sc["colors"])
t = [sc["transforms"].copy() for _ in range(2)]
t[1][:, 12] += 1e-4
for mode in ("resident", "animated"):
    for it in range(2):
        n = 5 if it == 0 else 60
        r.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            scene.tessellate(); frame.clear()
            if mode == "resident": scene.render(frame)
            else: scene.render(frame, t[i & 1], sc["colors"])
        r.synchronize(); dt = (time.perf_counter() - t0) / n
    print(mode, "%.3f ms/frame" % (dt * 1e3))
