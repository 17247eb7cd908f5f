"""GPU box: the marks of the three lanes over a few steps of new paths every frame (two Scenes in turn, FRAMES targets; CRH_TIMELINE=1 prints
them on kernel_times()). CRH_NO_OPTIMISTIC_UPLOAD=1: with the host's wait for the totals."""
import os
import sys

os.environ["CRH_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene

sc = scenes.scene_cubic_fill(10000)
r = Renderer(Configuration(1, 4, 4), device=0)
pair = [Scene(r, sc["batch"]), Scene(r, sc["batch"])]
frames = [Frame(r, 4096, 4096) for _ in range(int(os.environ.get("FRAMES", "2")))]


def loop(n):
    for i in range(n):
        k = i % 2
        pair[k] = Scene(r, sc["batch"], tessellate=False, existing=pair[k])
        pair[k].set_instances(sc["transforms"], sc["colors"])
        pair[k].tessellate()
        f = frames[i % len(frames)]
        f.clear()
        pair[k].render(f)


loop(40)
r.synchronize()
r.enable_timing(1)
loop(6)
r.synchronize()
r.kernel_times()
