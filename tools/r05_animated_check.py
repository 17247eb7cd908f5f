"""GPU box: what a moving scene costs and whether its frames are right. Two targets in turn, each consumed (crh_frame_synchronize) before it is
drawn into again. Modes: steady (resident instances) / same (set_instances with the SAME transforms) / moved (twenty zoomed views); every frame
of one cycle of the moved views is then compared with a fresh, synchronised render of its view."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene
w = sys.argv[1] if len(sys.argv) > 1 else "cubic"
sc = {"cubic": lambda: scenes.scene_cubic_fill(10000), "s100k": lambda: scenes.scene_cubic_fill(100000, (8192, 8192)), "glyphs": lambda: scenes.scene_glyphs(50000, (2048, 2048))}[w]()
W, H = sc["width"], sc["height"]
r = Renderer(Configuration(sc["msaa"], 4, 4), device=0)
scene = Scene(r, sc["batch"])
shown = [Frame(r, W, H), Frame(r, W, H)]
tr, co = sc["transforms"], sc["colors"]
scene.set_instances(tr, co)
n_sets = 20
zoom = [1.01 ** (k if k <= n_sets // 2 else n_sets - k) for k in range(n_sets)]
moved = []
for z in zoom:
    t = np.array(tr, dtype=np.float32, copy=True).reshape(-1, 16)
    t[:, [0, 1, 4, 5, 12, 13]] *= np.float32(z)
    moved.append(t)
def loop(n, mode, keep=None):
    for i in range(n):
        f = shown[i % 2]
        f.synchronize()
        if keep is not None and i >= 2: keep.append(f.download())  # what step i - 2 drew
        if mode == "same": scene.set_instances(tr, co)
        elif mode == "moved": scene.set_instances(moved[i % n_sets], co)
        scene.tessellate(); f.clear(); scene.render(f)
for mode in ("steady", "same", "moved", "steady", "moved"):
    if mode == "steady": scene.set_instances(tr, co)
    loop(60, mode); r.synchronize()
    t0 = time.perf_counter(); loop(40, mode); r.synchronize()
    print(w, mode, "%.4f ms/step" % ((time.perf_counter() - t0) / 40 * 1e3), flush=True)
if w != "s100k":
    frames = []
    loop(n_sets + 2, "moved", frames)
    r.synchronize()
    r2 = Renderer(Configuration(sc["msaa"], 4, 4), device=0)
    s2 = Scene(r2, sc["batch"]); f2 = Frame(r2, W, H)
    bad = 0
    for k, got in enumerate(frames):
        f2.clear(); s2.render(f2, moved[k % n_sets], co); r2.synchronize()
        bad += int(not np.array_equal(got, f2.download()))
    print("frames of one cycle that differ from a fresh render of their view:", bad, "of", len(frames))
