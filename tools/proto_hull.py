"""DEVELOPMENT TOOL (CPU): is there an exact lane-parallel form of convex_hull::andrew's chain walk? tools/proto_hull.cpp restates the walk
and three "pop round" formulations on the oracle's hull candidates; this runs them over the glyph scene (every glyph of the font at five
sizes), the cubic scenes and random path soups. Usage: python tools/proto_hull.py"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from contrast_renderer_amd import _ffi, scenes  # noqa: E402

SO = os.path.join(ROOT, "tools", "libproto_hull.so")


def build():
    src = os.path.join(ROOT, "tools", "proto_hull.cpp")
    deps = [src] + [os.path.join(ROOT, "oracle", f) for f in ("api.cpp", "tessellate.hpp", "ga.hpp", "curve.hpp", "raster.hpp")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-shared", "-I", os.path.join(ROOT, "include"), src, "-o", SO])
    lib = C.CDLL(SO)
    lib.oracle_tessellate.restype = C.c_void_p
    lib.oracle_tessellate.argtypes = [C.POINTER(_ffi.PathBatchC), C.c_int]
    lib.oracle_free.argtypes = [C.c_void_p]
    lib.proto_hull_compare.restype = C.c_long
    lib.proto_hull_compare.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_long)]
    lib.proto_hull_candidates.restype = C.c_uint32
    lib.proto_hull_candidates.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.c_uint32]
    return lib


def compare(lib, batch, label):
    handle = lib.oracle_tessellate(C.byref(batch.c), 8)
    out = {}
    for mode, name in ((0, "all at once"), (1, "later of two neighbours"), (2, "earlier of two neighbours")):
        stats = (C.c_long * 6)()
        lib.proto_hull_compare(handle, mode, stats)
        out[mode] = list(stats)
        print(f"{label}: pop rounds, {name}: {stats[1]} of {stats[0]} Shapes differ from the serial walk (first: Shape {stats[2]}); "
              f"{stats[3] / max(1, 2 * stats[0]):.1f} rounds per chain, {stats[4]} at most, {stats[5] / max(1, stats[0]):.0f} candidates per Shape")
    lib.oracle_free(handle)
    return out


def candidates(lib, batch, shape):
    handle = lib.oracle_tessellate(C.byref(batch.c), 1)
    buf = (C.c_float * 8192)()
    n = lib.proto_hull_candidates(handle, shape, buf, 4096)
    lib.oracle_free(handle)
    return np.frombuffer(buf, dtype=np.float32)[: 2 * n].reshape(-1, 2).copy()


def write_fixture(lib):
    """tests/golden/hull_pop_rounds_counterexample.json: the candidates of the first glyph Shape on which EVERY pop-round formulation leaves
    another hull than the serial walk (bit patterns of the f32 coordinates, emission order)."""
    import json
    sc = scenes.scene_glyphs(2000, (2048, 2048))
    differ = None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_hull_formulations import chains  # the numpy restatement the test itself uses
    for shape in range(2000):
        c = candidates(lib, sc["batch"], shape)
        if len(c) < 3 or len(c) > 16:
            continue
        serial, by_rounds = chains(c)
        if all(serial != r for r in by_rounds):
            differ = shape
            break
    assert differ is not None
    c = candidates(lib, sc["batch"], differ)
    path = os.path.join(ROOT, "tests", "golden", "hull_pop_rounds_counterexample.json")
    json.dump({"source": f"scenes.scene_glyphs(2000, (2048, 2048)), Shape {differ}: hull candidates in emission order (tools/proto_hull.py --fixture)",
               "candidates_f32_bits": [[int(np.float32(x).view(np.uint32)), int(np.float32(y).view(np.uint32))] for x, y in c]}, open(path, "w"), indent=1)
    print("wrote", path, "Shape", differ, len(c), "candidates")


if __name__ == "__main__":
    lib = build()
    if "--fixture" in sys.argv:
        write_fixture(lib)
        sys.exit(0)
    compare(lib, scenes.scene_glyphs(50000, (2048, 2048))["batch"], "50 000 glyphs")
    compare(lib, scenes.scene_cubic_fill(10000)["batch"], "10 000 cubic blobs")
    compare(lib, scenes.scene_mixed(2000, (1024, 1024), seed=3)["batch"], "2 000 mixed")
    compare(lib, scenes.scene_dashed_strokes(200)["batch"], "200 dashed strokes")
