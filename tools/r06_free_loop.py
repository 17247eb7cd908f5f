"""GPU box: the timed loop of bench.py by itself (tessellate + clear + render into ONE target, nothing consumed, up to three steps in flight) — for rocprofv3 timelines."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene
sc = scenes.scene_cubic_fill(10000, (4096, 4096), config_index=2)
r = Renderer(Configuration(1, 4, 4), device=0)
scene = Scene(r, sc["batch"]); frame = Frame(r, 4096, 4096)
scene.set_instances(sc["transforms"], sc["colors"])
def loop(n):
    for _ in range(n):
        scene.tessellate(); frame.clear(); scene.render(frame)
loop(60); r.synchronize(); loop(30); r.synchronize()
