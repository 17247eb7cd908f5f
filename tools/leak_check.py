import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from contrast_renderer_amd import scenes
from contrast_renderer_amd import renderer as R
sc = scenes.scene_mixed(12, (128,128), seed=2)
free0,_ = torch.cuda.mem_get_info()
for i in range(400):
    r = R.Renderer(R.Configuration(1 if i%2 else 4, 2, 4, 1), device=0)
    s = R.Scene(r, sc["batch"]); f = R.Frame(r, 128+i%64, 128); f.clear()
    if i % 4 == 0: f.keep_pass_state()  # (round 6: the frame's stencil / alpha / sample-colour planes are allocated — and have to go with the frame)
    s.render(f, sc["transforms"], sc["colors"])
    if i % 3 == 0: f.download()
    if i % 5 == 0: s2 = R.Scene(r, sc["batch"], existing=s); s = s2
    del f, s, r
    if i % 100 == 99:
        free,_ = torch.cuda.mem_get_info(); print(i, "free delta MB", (free0-free)/1e6)
