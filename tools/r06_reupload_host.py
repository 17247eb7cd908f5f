"""GPU box: host microseconds of every call of the `reupload` side block's step (bench.side_reupload), median over the timed steps, beside the step."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene
size = (4096, 4096)
sc = scenes.scene_cubic_fill(10000, size, config_index=2)
batch, transforms, colors = sc["batch"], sc["transforms"], sc["colors"]
renderer = Renderer(Configuration(msaa_sample_count=1, clip_nesting_counter_bits=4, winding_counter_bits=4), device=0)
scenes_ = [Scene(renderer, batch, tessellate=True), Scene(renderer, batch, tessellate=True)]
frames = [Frame(renderer, *size), Frame(renderer, *size)]
names = ["upload", "set_instances", "tessellate", "clear", "render"]
host = {n: [] for n in names}
def run(n):
    for i in range(n):
        k = i % 2
        t = [time.perf_counter()]
        scenes_[k] = Scene(renderer, batch, tessellate=False, existing=scenes_[k]); t.append(time.perf_counter())
        scenes_[k].set_instances(transforms, colors); t.append(time.perf_counter())
        scenes_[k].tessellate(); t.append(time.perf_counter())
        frames[k].clear(); t.append(time.perf_counter())
        scenes_[k].render(frames[k]); t.append(time.perf_counter())
        for j, nm in enumerate(names):
            host[nm].append(t[j + 1] - t[j])
run(60)
renderer.synchronize()
for v in host.values():
    del v[:]
t0 = time.perf_counter()
run(40)
t1 = time.perf_counter()
renderer.synchronize()
t2 = time.perf_counter()
print("step %.1f us (host loop alone %.1f us per step)" % ((t2 - t0) / 40 * 1e6, (t1 - t0) / 40 * 1e6))
for nm in names:
    v = sorted(host[nm])
    print("  %-14s median %7.1f us   max %7.1f" % (nm, v[len(v) // 2] * 1e6, v[-1] * 1e6))
