"""DEVELOPMENT TOOL (CPU): checks the boundary-edge + backdrop formulation (tools/proto_edges.cpp) against the oracle's triangle strips
on structured and unstructured scenes, bit for bit (pixels and final stencil bytes). Usage: python tools/proto_edges.py [n_seeds]"""
import ctypes as C
import math
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from contrast_renderer_amd import _ffi, batch_from_shapes, scenes  # noqa: E402

SO = os.path.join(ROOT, "tools", "libproto_edges.so")


def build():
    src = os.path.join(ROOT, "tools", "proto_edges.cpp")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-pthread", "-shared", src, "-o", SO])
    lib = C.CDLL(SO)
    lib.oracle_tessellate.restype = C.c_void_p
    lib.oracle_tessellate.argtypes = [C.POINTER(_ffi.PathBatchC), C.c_int]
    lib.oracle_status.argtypes = [C.c_void_p]
    lib.oracle_shape_status.argtypes = [C.c_void_p, C.c_uint32]
    lib.oracle_free.argtypes = [C.c_void_p]
    fp = C.POINTER(C.c_float)
    lib.proto_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, fp, fp, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_long)]
    return lib


def run(lib, batch, w, h, msaa, bits, t, c, label):
    handle = lib.oracle_tessellate(C.byref(batch.c), 8)
    n = batch.n_shapes
    t = np.ascontiguousarray(t, dtype=np.float32)
    c = np.ascontiguousarray(c, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    outs = []
    stats = (C.c_long * 2)()
    for mode in (0, 1, 2):
        img = np.zeros((h, w, 4), np.uint8)
        wind = np.zeros((h, w, msaa), np.uint8)
        lib.proto_render(handle, w, h, msaa, bits, t.ctypes.data_as(fp), c.ctypes.data_as(fp), 0, n, mode, img.ctypes.data, wind.ctypes.data, stats)
        outs.append((img, wind))
    lib.oracle_free(handle)
    dp = int((outs[0][0] != outs[1][0]).any(axis=2).sum())
    dw = int((outs[0][1] != outs[1][1]).sum())
    dp2 = int((outs[0][0] != outs[2][0]).any(axis=2).sum())  # the row-span form (mode 2) against the oracle's strips
    dw2 = int((outs[0][1] != outs[2][1]).sum())
    print(f"{label}: {n} shapes {w}x{h} msaa {msaa} bits {bits}: per sample: pixels differ {dp}, stencil bytes differ {dw}; "
          f"row spans: pixels differ {dp2}, stencil bytes differ {dw2}; tiles {stats[0]} pairs {stats[1]}")
    return dp + dw + dp2 + dw2


def good_shapes(lib, shapes):
    batch = batch_from_shapes(shapes)
    handle = lib.oracle_tessellate(C.byref(batch.c), 8)
    good = [s for s in range(len(shapes)) if lib.oracle_shape_status(handle, s) == 0]
    lib.oracle_free(handle)
    return batch_from_shapes([shapes[s] for s in good])


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    lib = build()
    bad = 0
    sc = scenes.scene_cubic_fill(300, (512, 512), r_lo=4.0, r_hi=64.0)
    bad += run(lib, sc["batch"], 512, 512, 1, 4, sc["transforms"], sc["colors"], "cubic")
    bad += run(lib, sc["batch"], 512, 512, 4, 4, sc["transforms"], sc["colors"], "cubic msaa4")
    sc = scenes.scene_quadratic(60, (512, 512))
    bad += run(lib, sc["batch"], 512, 512, 1, 4, sc["transforms"], sc["colors"], "quadratic")
    for seed in range(n_seeds):
        sc = scenes.scene_mixed(24, (256, 256), seed=seed)
        for msaa in (1, 4):
            bad += run(lib, sc["batch"], 200 + seed, 177, msaa, [1, 2, 4][seed % 3], sc["transforms"], sc["colors"], f"mixed {seed}")
    from test_gpu_fuzz import random_paths
    for seed in range(n_seeds):
        batch = good_shapes(lib, random_paths(300, 77 + seed))
        n = batch.n_shapes
        rng = np.random.RandomState(seed)
        t = scenes.place(256, 256, rng.uniform(0, 256, n), rng.uniform(0, 256, n), rng.uniform(5, 60, n))
        c = np.concatenate([rng.uniform(0, 1, (n, 3)), rng.uniform(0.2, 1, (n, 1))], axis=1).astype(np.float32)
        bad += run(lib, batch, 256, 256, 1 if seed % 2 == 0 else 4, 4, t, c, f"random paths {seed}")
        # extreme placements: slivers, huge coordinates
        t = scenes.place(256, 256, rng.uniform(-700, 900, n), rng.uniform(-700, 900, n), np.exp(rng.uniform(math.log(0.05), math.log(8000.0), n)))
        bad += run(lib, batch, 256, 256, 1, 4, t, c, f"random paths extreme {seed}")
    print("TOTAL differences:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())


def glyphs(n=600):
    lib = build()
    sc = scenes.scene_glyphs(n, (2048, 2048))
    return run(lib, sc["batch"], 2048, 2048, 1, 4, sc["transforms"], sc["colors"], f"glyphs {n}")
