"""Builds libcontrast_hip.so in-tree with hipcc for gfx950 (the .so travels to the GPU box with the snapshot).

-ffp-contract=off is part of the numerical contract: Rust never fuses a*b+c, and the parity tests compare bytes.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ("tessellate.hip", "raster.hip", "raster_edges.hip", "api.hip", "comm.hip", "text.cpp", "path.cpp")  # text.cpp / path.cpp: host-only (text.rs, path.rs:639-708)
HEADERS = ("ga.hpp", "fill.hpp", "stroke.hpp", "scene.hpp", "raster_params.hpp", "raster_common.hpp", "../../include/contrast_hip.h", "../../include/crh_fmath.h")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical"]
# Per-file flags. raster_edges.hip without LLVM's SLP vectorizer (round 6): it packs pairs of independent f32 operations of the per-sample code into
# v_pk_* instructions, whose operands are register PAIRS — lane-invariant values (sample positions, the tile's origin) end up duplicated in pairs that live
# for the whole kernel, and the raster kernels spilled exactly those (k_raster_edges<4,1,true,false>: 100 B of scratch per lane -> 12, 96 -> 95 registers;
# k_raster_fill 94 -> 76 registers; k_raster_rows 112 -> 95). Same results bit for bit (no contraction either way); S10k raster kernel 0.166 -> 0.151 ms alone,
# dashed scene 1.557 -> 1.481 (profiles/r06_experiments.txt). CRH_FILE_FLAGS="file.hip:-flag,-flag;..." replaces the table (A/B runs).
FILE_FLAGS = {"raster_edges.hip": ["-fno-slp-vectorize"], "raster.hip": ["-fno-slp-vectorize"], "tessellate.hip": ["-fno-slp-vectorize"]}  # (raster.hip: k_raster_tile<4,1,true,true> 150 -> 127 registers; tessellate.hip: k_tess_runs<true> 182 -> 146)
OUT = os.path.join(HERE, "libcontrast_hip.so")


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps if os.path.exists(d))


def file_flags():
    table = os.environ.get("CRH_FILE_FLAGS")
    if table is None:
        return FILE_FLAGS
    out = {}
    for entry in filter(None, table.split(";")):
        name, _, flags = entry.partition(":")
        out[name] = [f for f in flags.split(",") if f]
    return out


def build_library(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    objects = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", os.path.splitext(src)[0] + ".o")
        objects.append(obj)
        if not force and _newer(obj, [os.path.join(CSRC, src)] + headers):
            continue
        flags = (FLAGS + os.environ.get("CRH_EXTRA_FLAGS", "").split()) if src.endswith(".hip") else [f for f in FLAGS if not f.startswith("--offload-arch")]  # plain C++ for host-only files
        flags = flags + file_flags().get(src, [])
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, proc in procs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode(), file=sys.stderr)
    exports = write_export_map()
    if procs or not os.path.exists(OUT) or os.path.getmtime(exports) > os.path.getmtime(OUT):
        # librccl.so is dlopen()ed by comm.hip, not linked; the version script keeps the dynamic symbol table to the C ABI
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objects + ["-ldl", "-Wl,--version-script=" + exports, "-o", OUT]
        subprocess.check_call(cmd)
    return OUT


DEBUG_TAPS = ("crh_debug_frame_counters", "crh_debug_frame_counters16", "crh_debug_frame_words", "crh_debug_frame_bin_dump", "crh_debug_flat_batches", "crh_debug_pass_leaves_state")  # tools/ and one test read the raster kernels' counters through these


def declared_entry_points():
    """The names include/contrast_hip.h declares — the whole of the library's dynamic symbol table, with the debug taps."""
    import re
    with open(os.path.join(HERE, "..", "include", "contrast_hip.h")) as f:
        return sorted(set(re.findall(r"\b(crh_[a-z_0-9]+)\s*\(", f.read())) - {"crh_status"})


def write_export_map():
    """build/exports.map: a linker version script listing exactly the header's entry points. Everything else — the C++ internals shared by
    the translation units (crh::launch_*), the crh_internal_* accessors comm.hip uses, HIP's kernel stubs — stays local to the library."""
    path = os.path.join(HERE, "build", "exports.map")
    text = "{\n  global:\n" + "".join(f"    {name};\n" for name in declared_entry_points() + list(DEBUG_TAPS)) + "  local:\n    *;\n};\n"
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as f:
            f.write(text)
    return path


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
