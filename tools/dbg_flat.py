"""GPU box: k_bin_flat against k_bin_edges on tiny scenes (debugging aid): where do the images differ?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrast_renderer_amd import renderer as R, scenes
def draw(sc, itemwise):
    if itemwise: os.environ["CRH_BIN_ITEMWISE"] = "1"
    else: os.environ.pop("CRH_BIN_ITEMWISE", None)
    r = R.Renderer(R.Configuration(sc["msaa"], 4, sc["winding_bits"]), device=0)
    scene = R.Scene(r, sc["batch"]); f = R.Frame(r, sc["width"], sc["height"]); f.clear(); scene.render(f, sc["transforms"], sc["colors"])
    return f.download()
for n, size in ((1, (64, 64)), (2, (96, 96)), (5, (128, 128)), (40, (256, 256))):
    sc = scenes.scene_cubic_fill(n, size, r_lo=10.0, r_hi=30.0)
    a, b = draw(sc, False), draw(sc, True)
    d = (a != b).any(axis=2)
    ys, xs = np.nonzero(d)
    print(n, size, "differ:", d.sum(), "tiles:", sorted(set(zip((ys // 16).tolist(), (xs // 16).tolist())))[:12])
    if d.sum() and n == 1:
        print("flat alpha tile map (rows of tiles, 1 = any alpha):")
        for ty in range(size[1] // 16): print("".join("1" if a[ty*16:(ty+1)*16, tx*16:(tx+1)*16, 3].any() else "." for tx in range(size[0] // 16)), "  ", "".join("1" if b[ty*16:(ty+1)*16, tx*16:(tx+1)*16, 3].any() else "." for tx in range(size[0] // 16)))
