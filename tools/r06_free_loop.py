"""GPU box: the timed loop of bench.py by itself (tessellate + clear + render into ONE target, nothing consumed, up to three steps in flight) — for rocprofv3 timelines."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene
w = sys.argv[1] if len(sys.argv) > 1 else "cubic"
sc, size = {"cubic": lambda: (scenes.scene_cubic_fill(10000, (4096, 4096), config_index=2), 4096), "glyphs": lambda: (scenes.scene_glyphs(50000, (2048, 2048)), 2048),
            "dashed": lambda: (scenes.scene_dashed_strokes(2000, (4096, 4096)), 4096), "s100k": lambda: (scenes.scene_cubic_fill(100000, (8192, 8192), config_index=2), 8192)}[w]()
r = Renderer(Configuration(sc["msaa"], 4, 4), device=0)
scene = Scene(r, sc["batch"]); frame = Frame(r, size, size)
scene.set_instances(sc["transforms"], sc["colors"])
def loop(n):
    for _ in range(n):
        scene.tessellate(); frame.clear(); scene.render(frame)
loop(60); r.synchronize(); loop(30); r.synchronize()
