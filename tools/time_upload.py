"""DEVELOPMENT TOOL (GPU): crh_scene_upload of the metric's scene — the first one of a process, later new Scenes, re-uploads into an existing Scene."""
import time, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrast_renderer_amd import renderer as R, scenes
sc = scenes.scene_cubic_fill(10000)
r = R.Renderer(R.Configuration(1, 4, 4), 0)
ts = []
scs = []
for k in range(4):
    t = time.perf_counter(); s = R.Scene(r, sc["batch"], tessellate=False); ts.append(time.perf_counter() - t); scs.append(s)
print("new Scene uploads (ms):", [round(x * 1e3, 3) for x in ts])
s = scs[0]
ts = []
for k in range(4):
    t = time.perf_counter(); s = R.Scene(r, sc["batch"], tessellate=False, existing=s); ts.append(time.perf_counter() - t)
print("re-uploads (ms):", [round(x * 1e3, 3) for x in ts])
t = time.perf_counter(); s.tessellate(); r.synchronize(); print("first tessellate + sync (ms):", round((time.perf_counter() - t) * 1e3, 3))
t = time.perf_counter(); s.tessellate(); r.synchronize(); print("second tessellate + sync (ms):", round((time.perf_counter() - t) * 1e3, 3))
