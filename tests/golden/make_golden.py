"""Generates the committed golden fixtures (inputs + expected outputs) for the parity tests.

PROVENANCE: the expected outputs are produced by THIS REPOSITORY'S CPU oracle (oracle/, a restatement of the reference's
curve.rs / stroke.rs / fill.rs / convex_hull.rs / vertex.rs / renderer.rs / shaders.wgsl) — NOT by the reference itself,
which cannot be built or run here (no Rust toolchain, un-vendored geometric_algebra 0.3.0) and ships no golden vectors.
They pin the oracle against regressions and give the GPU tests a data-only target that travels to the GPU box.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from contrast_renderer_amd import scenes  # noqa: E402
from oracle import Oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def pack_batch(batch):
    import ctypes as C
    return dict(
        shape_path_begin=batch.shape_path_begin, path_segment_begin=batch.path_segment_begin, path_start=batch.path_start,
        path_stroke_options=batch.path_stroke_options, segment_types=batch.segment_types, control_data=batch.control_data,
        stroke_options=np.frombuffer(C.string_at(batch.stroke_options, C.sizeof(batch.stroke_options)), dtype=np.uint8).copy(),
        n_stroke_options=np.int64(batch.n_stroke_options), shape_dynamic_begin=batch.shape_dynamic_begin,
        dynamic_stroke_options=np.frombuffer(C.string_at(batch.dynamic_stroke_options, C.sizeof(batch.dynamic_stroke_options)), dtype=np.uint8).copy(),
        n_dynamic_stroke_options=np.int64(batch.n_dynamic_stroke_options))


def make(name, sc):
    oracle = Oracle(sc["batch"])
    assert oracle.status() == 0
    layout, vb, ib = oracle.all_shapes()
    image = oracle.render(sc["width"], sc["height"], sc["msaa"], sc["winding_bits"], sc["transforms"], sc["colors"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), layout=layout, vertex_bytes=vb, index_bytes=ib, image=image, transforms=sc["transforms"],
                        colors=sc["colors"], width=sc["width"], height=sc["height"], msaa=sc["msaa"], winding_bits=sc["winding_bits"], **pack_batch(sc["batch"]))
    print(name, "shapes", sc["batch"].n_shapes, "vertex bytes", vb.size, "index bytes", ib.size)


if __name__ == "__main__":
    make("mixed_24", scenes.scene_mixed(24, (192, 192)))
    make("quadratic_12", scenes.scene_quadratic(12, (192, 192)))
    make("cubic_fill_40", scenes.scene_cubic_fill(40, (192, 192), r_lo=8.0, r_hi=40.0))
    make("dashed_16", scenes.scene_dashed_strokes(16, (192, 192)))
