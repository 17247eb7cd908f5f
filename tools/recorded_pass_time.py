"""ms per frame of the 10k-path scene as a plain pass and as a recorded pass (crh_scene_render_draws: the OPS variant of the raster
kernel, host-side item merging, one readback per call) — GPU box."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from contrast_renderer_amd import scenes
from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene, RenderOperation as Op
sc = scenes.scene_cubic_fill(10000)
r = Renderer(Configuration(1, 4, 4), device=0)
scene = Scene(r, sc["batch"])
frame = Frame(r, 4096, 4096)
scene.set_instances(sc["transforms"], sc["colors"])
draws = np.array([d for i in range(10000) for d in ((i, i, int(Op.Stencil), 0, 0), (i, i, int(Op.Color), 0, 0))], dtype=np.uint32)
r.enable_timing(True)
for mode in ("plain", "recorded"):
    for it in range(2):
        n = 3 if it == 0 else 20
        r.synchronize(); r.kernel_times(); t0 = time.perf_counter()
        for i in range(n):
            frame.clear()
            if mode == "plain": scene.render(frame)
            else: scene.render_draws(frame, sc["transforms"], sc["colors"], draws)
        r.synchronize(); dt = (time.perf_counter() - t0) / n
    k = {}
    for name, ms, _ in r.kernel_times(): k.setdefault(name, []).append(ms)
    print(mode, "%.3f ms/frame" % (dt * 1e3), {n: round(float(np.mean(v)), 3) for n, v in k.items() if n.startswith("raster")})
