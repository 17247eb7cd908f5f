"""GPU box: one rank's step of the tile split — the whole scene tessellated and binned, one slab of tile rows drawn — kernel by kernel."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene, slab_rows
w = sys.argv[1] if len(sys.argv) > 1 else "cubic"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = scenes.scene_cubic_fill(10000) if w == "cubic" else scenes.scene_cubic_fill(100000, (8192, 8192))
r = Renderer(Configuration(1, 4, 4), device=0)
scene = Scene(r, sc["batch"]); scene.set_instances(sc["transforms"], sc["colors"])
for rank in ([None] + list(range(0, world, max(1, world // 4)))):
    f = Frame(r, sc["width"], sc["height"])
    if rank is not None: f.set_tile_rows(*slab_rows(sc["height"], rank, world))
    def loop(n):
        for _ in range(n):
            scene.tessellate(); f.clear(); scene.render(f)
    loop(60); r.synchronize()
    t0 = time.perf_counter(); loop(40); r.synchronize(); dt = (time.perf_counter() - t0) / 40
    r.enable_timing(1); loop(10); r.synchronize()
    agg = {}
    for name, ms, _ in r.kernel_times(): agg.setdefault(name, []).append(ms)
    r.enable_timing(False)
    print(w, "whole frame" if rank is None else f"slab {rank} of {world}", "%.4f ms/step" % (dt * 1e3), {k: round(sum(v) / len(v), 3) for k, v in agg.items() if k.startswith("raster")}, flush=True)
