"""Replays one seed of tests/test_gpu_fuzz.py::test_random_api_sequences_against_a_host_model with a log (debug tool)."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle; oracle.build()
from contrast_renderer_amd import scenes
from contrast_renderer_amd import renderer as R
from contrast_renderer_amd.renderer import RenderOperation as Op
from oracle.binding import Oracle, render_pass
seed = int(sys.argv[1]); eager_steps = set(int(v) for v in sys.argv[2].split(',')) if len(sys.argv) > 2 and sys.argv[2] != 'none' else set()
eager = False
rng = np.random.RandomState(4000 + seed)
msaa = int(rng.choice([1, 4]))
r = R.Renderer(R.Configuration(msaa, 2, 4, 2), device=0)
sizes = [(int(rng.randint(60, 200)), int(rng.randint(60, 200))) for _ in range(2)]
frames = [R.Frame(r, w, h) for w, h in sizes]
model = [np.zeros((h, w, 4), dtype=np.uint8) for w, h in sizes]
cleared = [True, True]
for f in frames: f.clear()
gpu_scenes, host = [None, None], [None, None]
def fresh_scene():
    while True:
        sc = scenes.scene_mixed(int(rng.randint(2, 25)), (160, 160), seed=int(rng.randint(0, 10000)))
        o = Oracle(sc["batch"])
        if o.status() == 0: return sc, o
for step in range(40):
    op = rng.randint(0, 7); k = int(rng.randint(0, 2))
    if op == 0 or gpu_scenes[k] is None:
        sc, o = fresh_scene()
        ex = rng.uniform() < 0.6
        gpu_scenes[k] = R.Scene(r, sc["batch"], existing=gpu_scenes[k] if ex else None)
        host[k] = dict(batch=sc["batch"], oracle=o, transforms=sc["transforms"], colors=sc["colors"], instances_set=False)
        print(step, "upload scene", k, "existing" if ex else "fresh", sc["batch"].n_shapes)
    elif op == 1:
        n = host[k]["batch"].n_shapes
        host[k]["transforms"] = scenes.place(160, 160, rng.uniform(0, 160, n), rng.uniform(0, 160, n), rng.uniform(5, 70, n))
        host[k]["colors"] = np.concatenate([rng.uniform(0, 1, (n, 3)), rng.uniform(0.2, 1, (n, 1))], axis=1).astype(np.float32)
        gpu_scenes[k].set_instances(host[k]["transforms"], host[k]["colors"]); host[k]["instances_set"] = True
        print(step, "set_instances", k)
    elif op in (2, 3, 4):
        j = int(rng.randint(0, 2)); w, h = sizes[j]
        if rng.uniform() < 0.6: frames[j].clear(); cleared[j] = True
        n = host[k]["batch"].n_shapes
        draws = [d for i in range(n) for d in ((i, i, int(Op.Stencil), 0, 0), (i, i, int(Op.Color), 0, 0))]
        kind = "plain"
        if op == 4 and n >= 3:
            inner = [d for i in range(1, n) for d in ((i, i, int(Op.Stencil), 1, 0), (i, i, int(Op.Color), 1, 0))]
            draws = [(0, 0, int(Op.Stencil), 0, 0), (0, 0, int(Op.Clip), 1, 0)] + inner + [(0, 0, int(Op.UnClip), 0, 0)]
            gpu_scenes[k].render_draws(frames[j], host[k]["transforms"], host[k]["colors"], draws); kind = "recorded"
        else:
            if not host[k]["instances_set"] or rng.uniform() < 0.5:
                gpu_scenes[k].render(frames[j], host[k]["transforms"], host[k]["colors"]); host[k]["instances_set"] = True; kind = "plain+set"
            else:
                gpu_scenes[k].render(frames[j]); kind = "resident"
        model[j], _ = render_pass(host[k]["oracle"], w, h, msaa, 4, 2, 2, host[k]["transforms"], host[k]["colors"], draws, load=None if cleared[j] else model[j])
        print(step, "pass", kind, "scene", k, "frame", j, "cleared" if cleared[j] else "over")
        cleared[j] = False
        if step in eager_steps:
            ok = np.array_equal(frames[j].download(), model[j]); print("   eager check", ok)
            if not ok: sys.exit(1)
    elif op == 5:
        pass
    else:
        j = int(rng.randint(0, 2))
        if not cleared[j]:
            ok = np.array_equal(frames[j].download(), model[j]); print(step, "download frame", j, ok)
            if not ok: sys.exit(1)
print("final", [np.array_equal(frames[j].download(), model[j]) for j in range(2) if not cleared[j]])
