"""Perspective instances of the oracle against mathematics: a pixel of a filled path seen through
perspective_projection * placement (utils.rs:181-203, main.rs:162-202) must be covered iff the ray through its centre hits the path's
plane in front of the eye, between the near and far planes, at a point with non-zero winding number with respect to the exact curve.
The ground truth inverts the homography in float64 and shares no code with the oracle's homogeneous rasterization."""
import math

import numpy as np
import pytest

from contrast_renderer_amd import Path, batch_from_shapes, utils
from test_oracle_ground_truth import flatten, pixel_centres, winding_numbers

SIZE = 96


def blob():
    p = Path(start=(0.9, 0.0))  # clockwise (y up), the reference's fill convention for cubic segments
    p.push_integral_cubic_curve((0.9, -0.6), (0.5, -0.9), (0.0, -0.8))
    p.push_rational_cubic_curve((1.0, 1.6, 0.7, 1.0), (-0.5, -0.7), (-1.0, -0.5), (-0.8, 0.1))
    p.push_integral_quadratic_curve((-0.7, 0.8), (-0.1, 0.7))
    p.push_rational_quadratic_curve(1.8, (0.5, 1.1), (0.6, 0.5))
    p.push_line((0.9, 0.0))
    return p


def camera(tilt, distance, aspect=1.0, near=1.0, far=100.0, shift=(0.0, 0.0)):
    projection = utils.perspective_projection(math.pi * 0.5, aspect, near, far)
    placement = utils.matrix_multiplication(utils.translation_matrix(shift[0], shift[1], distance), utils.rotation_matrix(tilt, (1.0, 0.0, 0.0)))
    return utils.matrix_multiplication(projection, placement)


def ground_truth(path, m, size, offsets):
    """-> [n_offsets, size*size] bool: covered at pixel centre + offset."""
    m = np.asarray(m, dtype=np.float64)
    # model (x, y, 1) -> (X, Y, W): screen = (X / W, Y / W)
    cx, cy, cz, cw = (np.array([m[r], m[4 + r], m[12 + r]]) for r in range(4))
    H = np.stack([(cx * 0.5 + cw * 0.5) * size, (cw * 0.5 - cy * 0.5) * size, cw])
    inverse = np.linalg.inv(H)
    polygon = flatten(path, 400)
    out = []
    for ox, oy in offsets:
        c = pixel_centres(size) + np.array([ox, oy])
        model = (inverse @ np.concatenate([c, np.ones((len(c), 1))], axis=1).T).T
        xy = model[:, :2] / model[:, 2:3]
        w = xy @ cw[:2] + cw[2]
        z = xy @ cz[:2] + cz[2]
        visible = (w > 0) & (z >= 0) & (z <= w)
        out.append(visible & (winding_numbers(polygon, xy) != 0))
    return np.stack(out)


CASES = {
    "tilted": dict(tilt=1.0, distance=2.2),
    "steep": dict(tilt=1.35, distance=1.6, shift=(0.2, -0.1)),
    "through_the_near_plane": dict(tilt=1.2, distance=1.25),     # part of the plane is closer than `near`: clipped per sample
    "behind_the_eye": dict(tilt=1.45, distance=0.45, near=0.05),  # part of the plane has w < 0: no clipping stage, the edge tests reject it
    "beyond_the_far_plane": dict(tilt=1.0, distance=2.2, far=2.3),
}


@pytest.mark.parametrize("case,msaa", [(c, 1) for c in sorted(CASES)] + [("behind_the_eye", 4), ("through_the_near_plane", 4)])
def test_perspective_fill_matches_the_unprojected_winding_number(oracle_lib, case, msaa):
    path = blob()
    m = camera(**CASES[case])
    batch = batch_from_shapes([([], [path])])
    oracle = oracle_lib.Oracle(batch)
    assert oracle.status() == 0
    image = oracle.render(SIZE, SIZE, msaa, 8, m.reshape(1, 16), np.array([[1.0, 1.0, 1.0, 1.0]], dtype=np.float32))
    alpha = image[..., 3].reshape(-1).astype(np.float64) / 255.0
    if msaa == 1:
        delta = 0.02
        truth = ground_truth(path, m, SIZE, [(0.0, 0.0), (delta, delta), (-delta, delta), (delta, -delta), (-delta, -delta)])
        sure = (truth == truth[0]).all(axis=0)  # the answer does not change within a hair of the centre
        assert (truth[0][sure] == (alpha[sure] > 0)).all(), f"{int((truth[0][sure] != (alpha[sure] > 0)).sum())} pixels differ away from the boundary"
        assert sure.mean() > 0.97 and 200 < truth[0].sum() < SIZE * SIZE - 200
    else:
        standard = [(6, 2), (14, 6), (2, 10), (10, 14)]
        delta = 0.02
        expected, sure = np.zeros(SIZE * SIZE), np.ones(SIZE * SIZE, dtype=bool)
        for x, y in standard:
            ox, oy = x / 16.0 - 0.5, y / 16.0 - 0.5
            truth = ground_truth(path, m, SIZE, [(ox, oy), (ox + delta, oy + delta), (ox - delta, oy + delta), (ox + delta, oy - delta), (ox - delta, oy - delta)])
            sure &= (truth == truth[0]).all(axis=0)
            expected += truth[0] / 4.0
        assert np.abs(expected - alpha)[sure].max() < 0.6 / 255.0  # the box-average resolve of exactly the covered standard sample positions
        assert sure.mean() > 0.9 and ((expected > 0) & (expected < 1) & sure).sum() > 40
