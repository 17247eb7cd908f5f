"""GPU box: tests/test_gpu_parity.py::test_binning_batches_by_cost_draw_the_same_frames under CRH_NO_DIRECT_LISTS=1, pass by pass."""
import os, sys
os.environ["CRH_NO_DIRECT_LISTS"] = "1"; os.environ["CRH_EDGE_PASS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from contrast_renderer_amd import scenes
from contrast_renderer_amd import renderer as gpu
from oracle.binding import Oracle
sc = scenes.scene_cubic_fill(700, (640, 512), r_lo=5.0, r_hi=70.0)
r = gpu.Renderer(gpu.Configuration(msaa_sample_count=sc["msaa"], winding_counter_bits=sc["winding_bits"]), device=0)
scene = gpu.Scene(r, sc["batch"]); oracle = Oracle(sc["batch"], 4)
frame = gpu.Frame(r, sc["width"], sc["height"])
moved = sc["transforms"].copy(); moved[:, 0] *= 3.0; moved[:, 5] *= 2.5; moved[:, 12] += 0.05
expect_of = {id(t): oracle.render(sc["width"], sc["height"], sc["msaa"], sc["winding_bits"], t, sc["colors"]) for t in (sc["transforms"], moved)}
for i, transforms in enumerate([sc["transforms"]] * 3 + [moved] * 20):
    frame.clear(); scene.render(frame, transforms, sc["colors"]); got = frame.download()
    d = (got != expect_of[id(transforms)]).any(axis=2)
    print("pass", i, "moved" if transforms is moved else "home", "differ", int(d.sum()), "nonzero got", int((got[..., 3] > 0).sum()), "nonzero expect", int((expect_of[id(transforms)][..., 3] > 0).sum()), flush=True)
