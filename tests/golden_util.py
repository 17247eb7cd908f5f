import ctypes as C
import glob
import os

import numpy as np

from contrast_renderer_amd import _ffi

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    n_so = int(z["n_stroke_options"])
    n_dyn = int(z["n_dynamic_stroke_options"])
    so = (_ffi.StrokeOptionsC * max(1, n_so)).from_buffer_copy(z["stroke_options"].tobytes())
    dyn = (_ffi.DynamicStrokeOptionsC * max(1, n_dyn)).from_buffer_copy(z["dynamic_stroke_options"].tobytes())
    batch = _ffi.PathBatch(z["shape_path_begin"], z["path_segment_begin"], z["path_start"], z["path_stroke_options"], z["segment_types"], z["control_data"],
                           [so[i] for i in range(n_so)], z["shape_dynamic_begin"], [dyn[i] for i in range(n_dyn)])
    return batch, z
