"""Host-side matrix / colour helpers of the reference's utils.rs that the render path is driven with (column-major 4x4 matrices as
flat [16] float32 arrays: element [4*c + r] = column c, row r — the layout of the instance buffer, shaders.wgsl:13-27)."""
import math

import numpy as np


def perspective_projection(field_of_view_y, aspect_ratio, near, far):
    """utils.rs:181-192: looks along +z, clip.w = z, depth 0 at `near` and 1 at `far`."""
    f32 = np.float32
    height = f32(1.0) / f32(math.tan(float(f32(field_of_view_y) * f32(0.5))))
    denominator = f32(1.0) / (f32(near) - f32(far))
    m = np.zeros(16, dtype=np.float32)
    m[0] = height / f32(aspect_ratio)
    m[5] = height
    m[10] = -f32(far) * denominator
    m[11] = 1.0
    m[14] = f32(near) * f32(far) * denominator
    return m


def matrix_multiplication(a, b):
    """utils.rs:194-203: the matrix product a * b of two column-major matrices, accumulated left to right in f32 like the reference."""
    a = np.asarray(a, dtype=np.float32).reshape(4, 4)  # a[c] = column c
    b = np.asarray(b, dtype=np.float32).reshape(4, 4)
    out = np.zeros((4, 4), dtype=np.float32)
    for c in range(4):
        acc = a[0] * b[c][0]
        for k in range(1, 4):
            acc = acc + a[k] * b[c][k]
        out[c] = acc
    return out.reshape(16)


def translation_matrix(x, y, z):
    """The matrix motor3d_to_mat4 (utils.rs:168-179) yields for a pure translator."""
    m = np.eye(4, dtype=np.float32).reshape(16)
    m[12], m[13], m[14] = x, y, z
    return m


def rotation_matrix(angle, axis):
    """The matrix motor3d_to_mat4 yields for rotate_around_axis(angle, axis) (utils.rs:143-146; axis of unit length), Rodrigues' form."""
    x, y, z = (float(v) for v in axis)
    c, s = math.cos(angle), math.sin(angle)
    t = 1.0 - c
    rows = [[t * x * x + c, t * x * y - s * z, t * x * z + s * y, 0.0],
            [t * x * y + s * z, t * y * y + c, t * y * z - s * x, 0.0],
            [t * x * z - s * y, t * y * z + s * x, t * z * z + c, 0.0],
            [0.0, 0.0, 0.0, 1.0]]
    return np.asarray(rows, dtype=np.float32).T.reshape(16).copy()


def srgb_to_linear(color):
    """utils.rs:205-215 (alpha untouched)."""
    out = np.array(color, dtype=np.float32)
    for i in range(3):
        out[i] = ((out[i] + np.float32(0.055)) / np.float32(1.055)) ** np.float32(2.4) if out[i] > np.float32(0.04045) else out[i] / np.float32(12.92)
    return out


def linear_to_srgb(color):
    """utils.rs:217-228 (alpha untouched)."""
    out = np.array(color, dtype=np.float32)
    for i in range(3):
        out[i] = out[i] ** np.float32(1.0 / 2.4) * np.float32(1.055) - np.float32(0.055) if out[i] > np.float32(0.0031308) else out[i] * np.float32(12.92)
    return out
