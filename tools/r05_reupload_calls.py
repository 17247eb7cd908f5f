"""GPU box: host time of every call of a re-upload step (new paths into the existing Scene every frame)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene
sc = scenes.scene_cubic_fill(10000)
r = Renderer(Configuration(1, 4, 4), device=0)
scene = Scene(r, sc["batch"]); scene.set_instances(sc["transforms"], sc["colors"])
other = Scene(r, sc["batch"])  # two Scenes in turn: an upload into the Scene of the frame in flight would wait for that frame
frame = Frame(r, 4096, 4096)
def loop(n, acc):
    global scene, other
    for i in range(n):
        scene, other = other, scene
        t0 = time.perf_counter(); scene = Scene(r, sc["batch"], tessellate=False, existing=scene)
        t1 = time.perf_counter(); scene.set_instances(sc["transforms"], sc["colors"])
        t2 = time.perf_counter(); scene.tessellate()
        t3 = time.perf_counter(); frame.clear(); scene.render(frame)
        t4 = time.perf_counter()
        for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)): acc[k] += d
loop(30, [0] * 4); r.synchronize()
acc = [0.0] * 4; t0 = time.perf_counter(); loop(40, acc); r.synchronize(); total = (time.perf_counter() - t0) / 40
print("%.4f ms/step; host ms: upload %.3f set_instances %.3f tessellate %.3f clear+render %.3f" % ((total * 1e3,) + tuple(a / 40 * 1e3 for a in acc)))
