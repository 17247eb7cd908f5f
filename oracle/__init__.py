"""CPU oracle — TEST INFRASTRUCTURE. Import only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

A C++ restatement of the reference's CPU tessellation (curve.rs, stroke.rs, fill.rs, convex_hull.rs, vertex.rs,
renderer.rs:20-60,121-141,177-215) and a software restatement of its stencil-then-cover passes (shaders.wgsl,
renderer.rs:267-355). PARITY UNPINNED against the real Rust crate: the reference has no tests or golden vectors and its
toolchain (cargo, geometric_algebra 0.3.0) is unavailable; the hand-derived KATs in tests/test_oracle_kat.py pin it.
"""
from .binding import Oracle, build, lib_path  # noqa: F401
