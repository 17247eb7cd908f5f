"""DEVELOPMENT TOOL (GPU, library built with -DCRH_ABLATE): average cycles a wavefront of k_bin_edges spends per phase."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRH_RASTER_DEBUG"] = "65536"
os.environ["CRH_EDGE_PASS"] = "1"
from contrast_renderer_amd import renderer as R, scenes
w = sys.argv[1] if len(sys.argv) > 1 else "cubic"
sc = {"cubic": lambda: scenes.scene_cubic_fill(10000), "glyphs": lambda: scenes.scene_glyphs(50000, (2048, 2048)), "dashed": lambda: scenes.scene_dashed_strokes(2000)}[w]()
size = {"cubic": 4096, "glyphs": 2048, "dashed": 4096}[w]
r = R.Renderer(R.Configuration(4 if w == "dashed" else 1, 4, 4), 0)
scene = R.Scene(r, sc["batch"]); scene.check(); scene.set_instances(sc["transforms"], sc["colors"])
frame = R.Frame(r, size, size)
out = (C.c_uint32 * 128)()
for _ in range(3):
    frame.clear(); scene.render(frame); r.synchronize()
    r.lib.crh_debug_frame_words(frame.handle, out)
n = scene.n_shapes
if not os.environ.get("CRH_BIN_ITEMWISE"):  # k_bin_flat: wavefront 0 of every workgroup
    names = ["item records + batch", "edge set-up (loads)", "triangle set-up + stores", "synthetic records + barrier", "rectangles, pool cleared", "pass 1 (edge table, row sums, the walk)", "pass 2", "pass 3", "folded hulls", "final flush"]
    vals = [out[80 + 2 * k] | (out[81 + 2 * k] << 32) for k in range(10)]
    ipg = int(os.environ.get("CRH_BIN_ITEMS", 0)) or min(32, max(1, (n + 1023) // 1024))
    groups = (n + ipg - 1) // ipg
    for name, v in zip(names, vals):
        print(f"{name:40s} {v / groups:10.0f} ticks per workgroup ({100.0 * v / max(1, sum(vals)):4.1f} %)")
    print(f"total {sum(vals) / groups:.0f} ticks per workgroup of {ipg} items, {groups} workgroups")
    longest, total, n_wg = out[120] | (out[121] << 32), out[122] | (out[123] << 32), out[124]
    if n_wg:
        print(f"workgroup lifetimes: {n_wg} workgroups, mean {total / n_wg:.0f} ticks, longest {longest} ticks = {longest * n_wg / max(1, total):.2f} x the mean")
    if os.environ.get("CRH_BIN_DUMP"):  # what every workgroup held, and a least-squares fit of its lifetime to it
        import numpy as np
        words = 2 * (n + 1) + 8 * n
        buf = (C.c_uint32 * words)()
        r.lib.crh_debug_frame_bin_dump.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
        assert r.lib.crh_debug_frame_bin_dump(frame.handle, buf, words) == 0
        rec = np.frombuffer(buf, dtype=np.uint32)[2 * (n + 1):].reshape(-1, 8)[:n_wg].astype(np.float64)
        names = ["life", "turns", "items", "tris", "edges", "cells", "edge rows", "longest tri walk"]
        print("per workgroup: " + ", ".join(f"{nm} mean {rec[:, k].mean():.0f} max {rec[:, k].max():.0f}" for k, nm in enumerate(names)))
        A = np.concatenate([np.ones((len(rec), 1)), rec[:, 1:]], axis=1)
        coef, *_ = np.linalg.lstsq(A, rec[:, 0], rcond=None)
        fit = A @ coef
        print("life ~ " + " + ".join(f"{c:.1f} x {nm}" for c, nm in zip(coef, ["1"] + names[1:])), f"(residual rms {np.sqrt(((fit - rec[:, 0]) ** 2).mean()):.0f} ticks)")
        for k in np.argsort(-rec[:, 0])[:8]:
            print("  longest:", {nm: int(v) for nm, v in zip(names, rec[k])})
        for k in np.argsort(rec[:, 0])[:4]:
            print("  shortest:", {nm: int(v) for nm, v in zip(names, rec[k])})
    sys.exit(0)
names = [["item data", "triangle set-up", "walk", "-", "-", "-", "final flush", "-"], ["item data + synth", "edge records", "rect + clear", "pass 1", "pass 2", "pass 3", "final flush", "-"]]
for wave in range(2):
    tot = 0
    for k in range(7):
        v = out[80 + 2 * (k + 8 * wave)] | (out[81 + 2 * (k + 8 * wave)] << 32)
        tot += v
        print(f"wave {wave} {names[wave][k]:18s} {v / n:10.0f} ticks per item")
    print(f"wave {wave} total {tot / n:.0f}")
