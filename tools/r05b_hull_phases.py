import sys, os, time
sys.path.insert(0, os.getcwd())
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Renderer, Scene
sc = scenes.scene_glyphs(50000, (2048, 2048))
r = Renderer(Configuration(1, 4, 4), device=0)
scene = Scene(r, sc["batch"])
for _ in range(5): scene.tessellate()
r.synchronize()
r.enable_timing(1)
for _ in range(10):
    scene.tessellate(); r.synchronize()
agg = {}
for name, ms, _ in r.kernel_times(): agg.setdefault(name, []).append(ms)
print({k: round(sum(v) / len(v), 4) for k, v in agg.items()})
