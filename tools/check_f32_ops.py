import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from contrast_renderer_amd import renderer as R
r = R.Renderer(R.Configuration(), 0)
rng = np.random.RandomState(0)
n = 1 << 20
a = (rng.uniform(0, 1, n) * 10.0 ** rng.uniform(-6, 3, n)).astype(np.float32)
b = (rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-3, 3, n)).astype(np.float32)
print("sqrt mismatches", int((r.selftest_fmath(6, a).view(np.uint32) != np.sqrt(a).view(np.uint32)).sum()))
print("div mismatches", int((r.selftest_fmath(7, a, b).view(np.uint32) != (a / b).view(np.uint32)).sum()))
ref = (np.float32(1.0) / np.sqrt(a * a + b * b)).astype(np.float32)
print("rsqrt-expr mismatches", int((r.selftest_fmath(8, a, b).view(np.uint32) != ref.view(np.uint32)).sum()))
ref = (a * b - np.float32(4.0) * a * b * b).astype(np.float32)
print("mul-sub expr mismatches", int((r.selftest_fmath(9, a, b).view(np.uint32) != ref.view(np.uint32)).sum()))
