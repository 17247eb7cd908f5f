"""GPU box: the marks of the three lanes over a few frames (CRH_TIMELINE=1 prints them on kernel_times()): two targets, each consumed before reuse;
steady (resident instances) / same (set_instances with the same transforms) / moved (zoomed views)."""
import sys, os
import numpy as np
os.environ["CRH_TIMELINE"] = "1"
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
def loop(n, mode):
    for i in range(n):
        f = shown[i % 2]; f.synchronize()
        scene.tessellate()
        if mode == "same": scene.set_instances(tr, co)
        if mode == "moved": scene.set_instances(moved[i % 20], co)
        f.clear(); scene.render(f)
for mode in sys.argv[1:] or ("steady", "same", "moved"):
    if mode == "steady": scene.set_instances(tr, co)
    loop(80, mode); r.synchronize()
    r.enable_timing(1); loop(8, mode); r.synchronize()
    sys.stderr.write(f"==== {mode}\n"); r.kernel_times(); r.enable_timing(False)
