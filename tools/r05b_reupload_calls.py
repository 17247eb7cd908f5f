"""GPU box: host time of every call of a re-upload step (new paths of the same structure into an existing Scene every frame; two Scenes
and two frames in turn, as bench.py --reupload). CRH_NO_OPTIMISTIC_UPLOAD=1: the way before round 5's second half (a wait for the totals)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene

sc = scenes.scene_cubic_fill(10000)
r = Renderer(Configuration(1, 4, 4), device=0)
pair = [Scene(r, sc["batch"]) for _ in range(int(os.environ.get("SCENES", "2")))]
frames = [Frame(r, 4096, 4096) for _ in range(int(os.environ.get("FRAMES", "2")))]
frames = frames * len(pair) if len(frames) == 1 else frames


def loop(n, acc):
    for i in range(n):
        k = i % len(pair)
        t0 = time.perf_counter()
        pair[k] = Scene(r, sc["batch"], tessellate=False, existing=pair[k])
        t1 = time.perf_counter()
        pair[k].set_instances(sc["transforms"], sc["colors"])
        t2 = time.perf_counter()
        pair[k].tessellate()
        t3 = time.perf_counter()
        frames[k].clear()
        t4 = time.perf_counter()
        pair[k].render(frames[k])
        t5 = time.perf_counter()
        for j, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[j] += d


loop(30, [0] * 5)
r.synchronize()
acc = [0.0] * 5
t0 = time.perf_counter()
loop(40, acc)
r.synchronize()
total = (time.perf_counter() - t0) / 40
print("%.4f ms/step; host ms: upload %.3f set_instances %.3f tessellate %.3f clear %.3f render %.3f" % ((total * 1e3,) + tuple(a / 40 * 1e3 for a in acc)))
