"""Debug helper (GPU box): per-category diff of the GPU tessellation against the oracle for a named scene."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from contrast_renderer_amd import scenes, renderer as R
from oracle import Oracle
from oracle.binding import split_shape, VERTEX_NAMES, VERTEX_SIZES

def run(name, sc):
    r = R.Renderer(R.Configuration(msaa_sample_count=sc["msaa"]), 0)
    scene = R.Scene(r, sc["batch"])
    o = Oracle(sc["batch"], 4)
    print(name, "status gpu", scene.status(), "oracle", o.status())
    nbad = 0
    for s in range(scene.n_shapes):
        gv = scene.shape(s); ov = o.shape(s)
        if not np.array_equal(gv[0], ov[0]) or not np.array_equal(gv[1], ov[1]):
            print(" shape", s, "layout differs", gv[0], ov[0], gv[1], ov[1]); nbad += 1; continue
        g = split_shape(*gv); w = split_shape(*ov)
        for cat, size in zip(VERTEX_NAMES, VERTEX_SIZES):
            if not np.array_equal(g[cat], w[cat]):
                a = g[cat].view(np.uint32).reshape(-1, size // 4); b = w[cat].view(np.uint32).reshape(-1, size // 4)
                rows, cols = np.nonzero(a != b)
                fa = a.view(np.float32); fb = b.view(np.float32)
                print(" shape", s, cat, "rows", len(set(rows)), "of", len(a), "cols", sorted(set(cols)), "first", rows[0], cols[0], fa[rows[0]], fb[rows[0]],
                      "max ulp", np.abs(a.astype(np.int64) - b.astype(np.int64))[rows, cols].max())
                nbad += 1
        for cat in ("line_indices", "joint_indices", "solid_indices"):
            if not np.array_equal(g[cat], w[cat]): print(" shape", s, cat, "differs"); nbad += 1
        if nbad > 12: break
    print(name, "bad entries", nbad)
    fr = R.Frame(r, sc["width"], sc["height"]); fr.clear(); scene.render(fr, sc["transforms"], sc["colors"]); img = fr.download()
    exp = o.render(sc["width"], sc["height"], sc["msaa"], sc["winding_bits"], sc["transforms"], sc["colors"])
    d = (img != exp).any(axis=2)
    print(name, "pixels differing", int(d.sum()), "of", d.size, "max abs", int(np.abs(img.astype(int) - exp.astype(int)).max()))
    if d.any():
        ys, xs = np.nonzero(d); print("  first", ys[:5], xs[:5], img[ys[0], xs[0]], exp[ys[0], xs[0]])

cases = {
    "cubic": lambda: scenes.scene_cubic_fill(40, (192, 192), r_lo=8.0, r_hi=40.0),
    "quad": lambda: scenes.scene_quadratic(12, (192, 192)),
    "dash": lambda: scenes.scene_dashed_strokes(16, (192, 192)),
    "mixed": lambda: scenes.scene_mixed(24, (192, 192)),
}
for name in (sys.argv[1:] or cases):
    run(name, cases[name]())
