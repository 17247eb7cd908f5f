"""GPU box: which of the per-shard layers differ from the oracle's render of the same shard? (debugging aid)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contrast_renderer_amd import renderer as R, scenes, distributed as D
from oracle.binding import Oracle
seed, world, size, n = 25, 4, (320, 256), 60
sc = scenes.scene_mixed(n, size, seed=seed)
delay = len(sys.argv) > 1
o = Oracle(sc["batch"])
r = R.Renderer(R.Configuration(1, 4, 4), device=0)
layers16, layers8, keep, wants = [], [], [], []
for rank in range(world):
    b, e = D.shard_range(n, rank, world)
    if delay:
        o.render(size[0], size[1], 1, 4, sc["transforms"], sc["colors"], b, e)
    scene = R.Scene(r, sc["batch"].slice_shapes(b, e))
    keep.append(scene)
    for fmt, out in ((R.FORMAT_RGBA16F, layers16), (R.FORMAT_RGBA8, layers8)):
        frame = R.Frame(r, *size, fmt)
        frame.clear()
        scene.render(frame, sc["transforms"][b:e], sc["colors"][b:e])
        out.append(frame)
for rank in range(world):
    b, e = D.shard_range(n, rank, world)
    wants.append(o.render(size[0], size[1], 1, 4, sc["transforms"], sc["colors"], b, e))
whole = o.render(size[0], size[1], 1, 4, sc["transforms"], sc["colors"])
b8 = [f.download() for f in layers8]
for k in range(world):
    d = np.abs(b8[k].astype(int) - wants[k].astype(int)).max(axis=2)
    print("layer8", k, "max", d.max(), "px", (d > 0).sum())
h16 = [f.download() for f in layers16]
for k in range(world):
    d = np.abs(h16[k].astype(np.float64) * 255 - wants[k]).max(axis=2)
    print("layer16", k, "max", d.max(), "px", (d > 0.6).sum())
comms = [R.Comm(r, 0, world)]
comms += [R.Comm(r, k, world, rank0=comms[0]) for k in range(1, world)]
result = R.Frame(r, *size)
for name, layers in (("16F", layers16), ("8", layers8), ("16F", layers16)):
    comms[0].local_exchange(layers, result)
    image = result.download()
    d = np.abs(image.astype(int) - whole.astype(int)).max(axis=2)
    print(name, "image vs whole: max", d.max(), "px>1", (d > 1).sum())
