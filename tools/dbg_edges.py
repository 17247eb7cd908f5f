"""DEVELOPMENT TOOL (GPU): compares the edge pass with the triangle pass (CRH_TRIANGLE_PASS=1) on a scene and bisects the first Shape
whose prefix render differs. Usage: python tools/dbg_edges.py glyphs 600"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from contrast_renderer_amd import renderer as R  # noqa: E402
from contrast_renderer_amd import scenes  # noqa: E402


def render(batch, t, c, w, h, msaa, triangle):
    if triangle:
        os.environ["CRH_TRIANGLE_PASS"] = "1"
    else:
        os.environ.pop("CRH_TRIANGLE_PASS", None)
    r = R.Renderer(R.Configuration(msaa, 4, 4), device=0)
    scene = R.Scene(r, batch)
    assert scene.status() == 0
    frame = R.Frame(r, w, h)
    frame.clear()
    scene.render(frame, t, c)
    return frame.download()


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "glyphs"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    if kind == "glyphs":
        sc = scenes.scene_glyphs(n, (2048, 2048))
        w = h = 2048
    else:
        sc = scenes.scene_cubic_fill(n, (1024, 1024))
        w = h = 1024
    batch, t, c = sc["batch"], np.asarray(sc["transforms"], np.float32), np.asarray(sc["colors"], np.float32)
    a = render(batch, t, c, w, h, 1, False)
    b = render(batch, t, c, w, h, 1, True)
    diff = (a != b).any(axis=2)
    print("shapes", batch.n_shapes, "differing pixels", int(diff.sum()))
    if not diff.any():
        return
    ys, xs = np.nonzero(diff)
    print("first differing pixels", list(zip(xs[:10].tolist(), ys[:10].tolist())))
    lo, hi = 0, batch.n_shapes  # smallest prefix length that differs
    while hi - lo > 1:
        mid = (lo + hi) // 2
        sub = batch.slice_shapes(0, mid)
        d = (render(sub, t[:mid], c[:mid], w, h, 1, False) != render(sub, t[:mid], c[:mid], w, h, 1, True)).any()
        if d:
            hi = mid
        else:
            lo = mid
    s = hi - 1
    print("first bad shape", s)
    one = batch.slice_shapes(s, s + 1)
    a1 = render(one, t[s:s + 1], c[s:s + 1], w, h, 1, False)
    b1 = render(one, t[s:s + 1], c[s:s + 1], w, h, 1, True)
    d1 = (a1 != b1).any(axis=2)
    ys, xs = np.nonzero(d1)
    print("alone: differing", int(d1.sum()), "bbox", (xs.min(), xs.max(), ys.min(), ys.max()) if d1.any() else None)
    print("paths", one.n_paths, "segments", one.n_segments)
    scene = R.Scene(R.Renderer(R.Configuration(1, 4, 4), device=0), one)
    layout, vb, ib = scene.all_shapes()
    print("layout", layout)
    if d1.any():
        x0, x1, y0, y1 = xs.min(), xs.max(), ys.min(), ys.max()
        print("edges alpha\n", a1[y0:y1 + 1, x0:x1 + 1, 3])
        print("tris alpha\n", b1[y0:y1 + 1, x0:x1 + 1, 3])


if __name__ == "__main__":
    main()
