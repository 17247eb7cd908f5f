"""CPU tests: the oracle still reproduces the committed golden fixtures, and the C-ABI library loads and exports every
symbol include/contrast_hip.h declares (no compute calls: there is no GPU here)."""
import os
import re

import numpy as np
import pytest

from golden_util import golden_names, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", golden_names())
def test_oracle_reproduces_golden(oracle_lib, name):
    batch, z = load_golden(name)
    o = oracle_lib.Oracle(batch)
    assert o.status() == 0
    layout, vb, ib = o.all_shapes()
    assert np.array_equal(layout, z["layout"]) and np.array_equal(vb, z["vertex_bytes"]) and np.array_equal(ib, z["index_bytes"])
    image = o.render(int(z["width"]), int(z["height"]), int(z["msaa"]), int(z["winding_bits"]), z["transforms"], z["colors"])
    assert np.array_equal(image, z["image"])


def test_golden_fixtures_exist():
    assert len(golden_names()) >= 4


def test_c_abi_exports_every_declared_symbol():
    """The library must build (hipcc cross-compiles gfx950 without a GPU) and export exactly the header's entry points."""
    import __graft_entry__ as entry
    entry.build()
    from contrast_renderer_amd import _ffi
    lib = _ffi.load_library()
    header = open(os.path.join(ROOT, "include", "contrast_hip.h")).read()
    declared = set(re.findall(r"\b(crh_[a-z_0-9]+)\s*\(", header))
    declared -= {"crh_status"}
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in contrast_hip.h but not exported by libcontrast_hip.so"
    assert set(lib._crh_signatures) == declared, set(lib._crh_signatures) ^ declared
    assert b"gfx950" in lib.crh_version()
    # ... and nothing else: the dynamic symbol table is the C ABI (plus three debug taps of tools/), no C++ internals
    import subprocess
    from contrast_renderer_amd import build as b
    table = subprocess.run(["nm", "-D", "--defined-only", b.OUT], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in table.splitlines() if line.strip()}
    assert exported == declared | set(b.DEBUG_TAPS), sorted(exported ^ (declared | set(b.DEBUG_TAPS)))


def test_product_has_no_cpu_fallback_and_never_touches_the_oracle():
    """The product path must fail loudly without the HIP device, and nothing under contrast_renderer_amd/ may reference oracle/."""
    import subprocess
    pkg = os.path.join(ROOT, "contrast_renderer_amd")
    hits = subprocess.run(["grep", "-rnE", r"^\s*(from|import)\s+oracle|#include\s+\"[./]*oracle", pkg, "--include=*.py", "--include=*.hip", "--include=*.hpp"],
                          capture_output=True, text=True).stdout
    assert hits == "", hits
    import torch
    if not torch.cuda.is_available():
        from contrast_renderer_amd import ContrastError
        from contrast_renderer_amd.renderer import Renderer
        with pytest.raises(ContrastError):
            Renderer()


def test_renderer_new_validates_stencil_bits_like_the_reference():
    """renderer.rs:433-435 — checked before any device is touched."""
    import ctypes as C
    from contrast_renderer_amd import _ffi
    lib = _ffi.load_library()
    handle = C.c_void_p()
    for winding, clip in ((0, 4), (5, 4), (8, 1)):
        cfg = _ffi.ConfigC(1, clip, winding, 0)
        assert lib.crh_renderer_create(C.byref(cfg), 0, C.byref(handle)) == _ffi.ERR_NUMBER_OF_STENCIL_BITS_IS_UNSUPPORTED


def test_descriptor_conversion_matches_oracle(oracle_lib):
    """crh_convert_dynamic_stroke_options is pure host code (renderer.rs:29-60): compare with the oracle's restatement."""
    import ctypes as C
    from contrast_renderer_amd import Cap, DashInterval, DynamicStrokeOptions, Join, _ffi
    from oracle.binding import _load
    lib, olib = _ffi.load_library(), _load()
    rng = np.random.RandomState(3)
    for _ in range(200):
        if rng.rand() < 0.5:
            n = rng.randint(1, 7)
            pattern = [DashInterval(rng.rand(), rng.rand() + 1, Cap(rng.randint(7)), Cap(rng.randint(7))) for _ in range(n)]
            o = DynamicStrokeOptions.Dashed(Join(rng.randint(3)), pattern, rng.randn()).to_c()
        else:
            o = DynamicStrokeOptions.Solid(Join(rng.randint(3)), Cap(rng.randint(7)), Cap(rng.randint(7))).to_c()
        a, b = _ffi.DynamicStrokeDescriptorC(), _ffi.DynamicStrokeDescriptorC()
        ra, rb = lib.crh_convert_dynamic_stroke_options(C.byref(o), C.byref(a)), olib.oracle_convert_dynamic_stroke_options(C.byref(o), C.byref(b))
        assert ra == rb
        if ra == 0:
            assert bytes(a) == bytes(b)


def test_header_is_plain_c():
    """include/contrast_hip.h is the FFI boundary: it must compile as C99 / C11 (what bindgen, cgo or a JNI shim would feed on), with
    warnings as errors."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "abi.c")
        with open(src, "w") as f:
            f.write('#include "contrast_hip.h"\nint main(void) { crh_config c = {1, 4, 4, 0, 0, 0, 0}; crh_draw d = {0, 0, CRH_OP_STENCIL, 0, 0}; (void)c; (void)d; return 0; }\n')
        for std in ("c99", "c11"):
            run = subprocess.run(["gcc", f"-std={std}", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), "-fsyntax-only", src],
                                 capture_output=True, text=True)
            assert run.returncode == 0, run.stderr


def test_binning_runs_partition_the_items_within_a_batch_and_start_with_the_long_ones(monkeypatch):
    """Host logic of the binning kernel's batches by cost (csrc/raster_edges.hip flat_batches; no device involved): whatever the items cost,
    the runs partition 0 .. n in order, every run fits ONE batch (items, triangles, edges, tile cells), an item wider than the pool is a run
    of its own, an item that is not binned there takes a place but nothing else, and the runs come longest predicted life first."""
    import ctypes as C
    import numpy as np
    import __graft_entry__ as entry
    entry.build()
    from contrast_renderer_amd import _ffi
    lib = _ffi.load_library()
    lib.crh_debug_flat_batches.restype = C.c_int
    lib.crh_debug_flat_batches.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
    for var in ("CRH_BIN_BATCH_TICKS", "CRH_BIN_BATCH_ITEMS", "CRH_BIN_BATCH_ORDER"):
        monkeypatch.delenv(var, raising=False)
    rng = np.random.default_rng(11)
    limits = (C.c_uint32 * 4)()
    lib.crh_debug_flat_batches(None, 0, np.zeros((1, 2), dtype=np.uint32).ctypes.data_as(C.POINTER(C.c_uint32)), 1, limits)
    pool = limits[3]  # (the tables scale with the kernel's workgroup shape: an item's rectangle beyond the pool is reported as "too wide", never as a count)
    for n in (0, 1, 31, 33, 1000, 20000):
        tris = rng.integers(0, 60, n).astype(np.uint32)
        edges = rng.integers(0, 200, n).astype(np.uint32)
        cells = np.minimum(rng.lognormal(3.0, 1.5, n), pool - pool // 10).astype(np.uint32)
        folded = (rng.uniform(size=n) < 0.1).astype(np.uint32)
        kind = rng.uniform(size=n)  # 3 %: wider than the pool, 3 %: not binned by this kernel
        cost = np.zeros((n, 2), dtype=np.uint32)
        cost[:, 0] = np.where(kind < 0.03, 0xFFFFFFFF, np.where(kind < 0.06, 0, cells))
        cost[:, 1] = np.where(kind < 0.06, 0x80000000, tris | (edges << 9) | (folded << 29))
        runs = np.zeros((max(1, n), 2), dtype=np.uint32)
        limits = (C.c_uint32 * 4)()
        k = lib.crh_debug_flat_batches(cost.ctypes.data_as(C.POINTER(C.c_uint32)), n, runs.ctypes.data_as(C.POINTER(C.c_uint32)), max(1, n), limits)
        assert k >= 0 and (k == 0) == (n == 0)
        max_items, max_tris, max_edges, max_cells = limits[:]
        runs = runs[:k]
        in_order = runs[np.argsort(runs[:, 0], kind="stable")]
        assert n == 0 or (in_order[0, 0] == 0 and in_order[-1, 1] == n and (in_order[1:, 0] == in_order[:-1, 1]).all() and (runs[:, 1] > runs[:, 0]).all())
        predicted = []
        for a, b in runs:
            c = cost[a:b]
            alone = c[:, 0] == 0xFFFFFFFF
            skipped = (c[:, 1] >> 31) != 0
            assert not alone.any() or b - a == 1, "an item wider than the pool shares a run"
            t = np.where(skipped, 0, c[:, 1] & 0x1FF).sum()
            e = np.where(skipped, 0, (c[:, 1] >> 9) & 0x3FF).sum()
            cl = np.where(alone, 0, c[:, 0]).astype(np.uint64)
            f = np.where(skipped, 0, (c[:, 1] >> 29) & 1).sum()
            assert b - a <= max_items and t <= max_tris and e <= max_edges and cl.sum() <= max_cells
            predicted.append(76000.0 + 2000.0 * (b - a) + 40.0 * t + 250.0 * e + 60.0 * float(cl.sum()) + 45.0 * float(cl.max()) + 8000.0 * f)
        assert all(x >= y - 1.0 for x, y in zip(predicted, predicted[1:])), "the runs are not in falling order of their predicted life"
