"""contrast_renderer_amd — MI355X-native (gfx950, HIP) tessellate + tile-raster hot path behind
contrast_renderer's Path / Shape / Renderer API. See DESIGN.md and include/contrast_hip.h.

The product is libcontrast_hip.so (hand-written HIP kernels + C ABI). This package is the host-side mirror of
the reference's interface; it has no CPU fallback and never imports the oracle.
"""
from ._ffi import ContrastError, PathBatch  # noqa: F401
from .path import (Cap, CurveApproximation, DashInterval, DynamicStrokeOptions, Join, Path, SegmentType, StrokeOptions,  # noqa: F401
                   batch_from_shapes)

__all__ = ["ContrastError", "PathBatch", "Cap", "CurveApproximation", "DashInterval", "DynamicStrokeOptions", "Join", "Path", "SegmentType",
           "StrokeOptions", "batch_from_shapes"]
