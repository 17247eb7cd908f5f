"""GPU box: the steady loop of bench.py (tessellate + clear + render, nothing consumed) into ONE target or into TWO in turn — ms per step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene
sc = scenes.scene_cubic_fill(10000, (4096, 4096), config_index=2)
r = Renderer(Configuration(1, 4, 4), device=0)
scene = Scene(r, sc["batch"]); frames = [Frame(r, 4096, 4096), Frame(r, 4096, 4096)]
scene.set_instances(sc["transforms"], sc["colors"])
def loop(n, targets):
    for i in range(n):
        f = frames[i % targets]
        scene.tessellate(); f.clear(); scene.render(f)
for targets in (1, 2, 1, 2):
    loop(60, targets); r.synchronize()
    best = []
    for _ in range(5):
        t0 = time.perf_counter(); loop(20, targets); r.synchronize(); best.append((time.perf_counter() - t0) / 20 * 1e3)
    print("targets", targets, " ".join("%.4f" % b for b in best))
