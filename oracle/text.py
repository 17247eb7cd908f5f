"""oracle/text.py — TEST INFRASTRUCTURE (never imported by the product): an independent pure-Python restatement of the glyph producer,
text.rs (paths_of_glyph :97-104, calculate_aligned_positions! :145-230, paths_of_text :236-263) + Path::transform (path.rs:387-439),
and of what text.rs observes of the un-vendored crate ttf-parser 0.14.0 (Cargo.toml:20).

PARITY UNPINNED: the reference has no tests or fixtures for this path and cannot be built here. This file is written from the
published TrueType table formats (glyf / loca / cmap / hmtx / hhea / OS/2 / kern), in a different style from the product's
csrc/text.cpp (whole-contour point lists here, a streaming state machine there), so that agreement between the two is evidence about
both; the outline walk conventions that the crate chooses (start point of a contour that begins off-curve, the explicit closing
segment before close()) are RECALLED, see SURVEY.md Appendix D.
"""
import struct

import numpy as np

F = np.float32
SEGMENT_FLOATS = (2, 4, 6, 5, 10)


class Face:
    def __init__(self, data: bytes):
        self.data = data
        base = 0
        if data[:4] == b"ttcf":
            base = struct.unpack(">I", data[12:16])[0]
        n_tables = struct.unpack(">H", data[base + 4:base + 6])[0]
        self.tables = {}
        for i in range(n_tables):
            tag, _, off, length = struct.unpack(">4sIII", data[base + 12 + 16 * i:base + 28 + 16 * i])
            self.tables[tag] = data[off:off + length]
        head, hhea, maxp = self.tables[b"head"], self.tables[b"hhea"], self.tables[b"maxp"]
        self.units_per_em = struct.unpack(">H", head[18:20])[0]
        self.loc_format = struct.unpack(">h", head[50:52])[0]
        self.n_glyphs = struct.unpack(">H", maxp[4:6])[0]
        self.n_h_metrics = struct.unpack(">H", hhea[34:36])[0]
        os2 = self.tables.get(b"OS/2", b"")
        typo = len(os2) >= 78 and (struct.unpack(">H", os2[62:64])[0] & 0x80) != 0
        if typo:
            self.ascender, self.descender, self.line_gap = struct.unpack(">hhh", os2[68:74])
        else:
            self.ascender, self.descender, self.line_gap = struct.unpack(">hhh", hhea[4:10])
        self.height = self.ascender - self.descender
        self.x_height = struct.unpack(">h", os2[86:88])[0] if len(os2) >= 88 and struct.unpack(">H", os2[0:2])[0] >= 2 else None
        loca = self.tables.get(b"loca", b"")
        if self.loc_format == 0:
            self.loca = [2 * v for v in struct.unpack(f">{len(loca) // 2}H", loca[:len(loca) // 2 * 2])]
        else:
            self.loca = list(struct.unpack(f">{len(loca) // 4}I", loca[:len(loca) // 4 * 4]))
        self._cmap = self._unicode_cmap()
        self.kern = self._kern_pairs()

    # ---- cmap: all Unicode subtables decoded into one dictionary (first subtable that maps a code point wins)
    def _unicode_cmap(self):
        cmap = self.tables.get(b"cmap", b"")
        out = {}
        if len(cmap) < 4:
            return out
        n = struct.unpack(">H", cmap[2:4])[0]
        for i in range(n):
            platform, encoding, off = struct.unpack(">HHI", cmap[4 + 8 * i:12 + 8 * i])
            sub = cmap[off:]
            fmt = struct.unpack(">H", sub[0:2])[0]
            if not (platform == 0 or (platform == 3 and (encoding == 1 or (encoding == 10 and fmt in (12, 13))))):
                continue
            mapping = {}
            if fmt == 4:
                seg = struct.unpack(">H", sub[6:8])[0] // 2
                ends = struct.unpack(f">{seg}H", sub[14:14 + 2 * seg])
                starts = struct.unpack(f">{seg}H", sub[16 + 2 * seg:16 + 4 * seg])
                deltas = struct.unpack(f">{seg}h", sub[16 + 4 * seg:16 + 6 * seg])
                ro_at = 16 + 6 * seg
                offsets = struct.unpack(f">{seg}H", sub[ro_at:ro_at + 2 * seg])
                for s in range(seg):
                    for code in range(starts[s], ends[s] + 1):
                        if code in mapping:
                            continue
                        if offsets[s] == 0:
                            mapping[code] = (code + deltas[s]) & 0xFFFF
                        elif offsets[s] != 0xFFFF:
                            at = ro_at + 2 * s + offsets[s] + 2 * (code - starts[s])
                            if at + 2 <= len(sub):
                                value = struct.unpack(">H", sub[at:at + 2])[0]
                                if value != 0:
                                    gid = (value + deltas[s]) & 0xFFFF
                                    if gid < 0x8000:
                                        mapping[code] = gid
            elif fmt == 12:
                groups = struct.unpack(">I", sub[12:16])[0]
                for g in range(groups):
                    start, end, gid = struct.unpack(">III", sub[16 + 12 * g:28 + 12 * g])
                    for code in range(start, end + 1):
                        if gid + code - start <= 0xFFFF:
                            mapping.setdefault(code, gid + code - start)
            elif fmt == 0:
                for code in range(256):
                    if sub[6 + code]:
                        mapping[code] = sub[6 + code]
            elif fmt == 6:
                first, count = struct.unpack(">HH", sub[6:10])
                for k in range(count):
                    mapping[first + k] = struct.unpack(">H", sub[10 + 2 * k:12 + 2 * k])[0]
            for code, gid in mapping.items():
                out.setdefault(code, gid)
        return out

    def _kern_pairs(self):
        kern = self.tables.get(b"kern", b"")
        if len(kern) < 10 or struct.unpack(">HH", kern[0:4]) [0] != 0 or struct.unpack(">H", kern[2:4])[0] == 0:
            return None
        _, length, coverage = struct.unpack(">HHH", kern[4:10])
        if not (coverage & 1) or (coverage >> 8) != 0:
            return None
        n_pairs = struct.unpack(">H", kern[10:12])[0]
        pairs = {}
        for i in range(n_pairs):
            left, right, value = struct.unpack(">HHh", kern[18 + 6 * i:24 + 6 * i])
            pairs[(left, right)] = value
        return pairs

    def glyph_index(self, code):
        return self._cmap.get(code)

    def hor_advance(self, gid):
        hmtx = self.tables.get(b"hmtx", b"")
        if gid >= self.n_glyphs or self.n_h_metrics == 0:
            return None
        i = min(gid, self.n_h_metrics - 1)
        return struct.unpack(">H", hmtx[4 * i:4 * i + 2])[0]

    def glyph_data(self, gid):
        if gid >= self.n_glyphs or gid + 1 >= len(self.loca):
            return None
        a, b = self.loca[gid], self.loca[gid + 1]
        glyf = self.tables.get(b"glyf", b"")
        return glyf[a:b] if a < b <= len(glyf) else None

    def bounding_box(self, gid):
        g = self.glyph_data(gid)
        return struct.unpack(">hhhh", g[2:10]) if g and len(g) >= 10 else None

    # ---- outline: contours of (x, y, on_curve) under an affine transform [a b c d e f]
    def contours(self, gid, transform=None, depth=0):
        g = self.glyph_data(gid)
        if not g or len(g) < 10 or depth >= 32:
            return []
        n_contours = struct.unpack(">h", g[0:2])[0]
        if n_contours > 0:
            ends = struct.unpack(f">{n_contours}H", g[10:10 + 2 * n_contours])
            n_points = ends[-1] + 1
            if n_points == 1:
                return []
            at = 10 + 2 * n_contours
            at += 2 + struct.unpack(">H", g[at:at + 2])[0]
            flags = []
            while len(flags) < n_points:
                flag = g[at]
                at += 1
                count = 1
                if flag & 8:
                    count += g[at]
                    at += 1
                flags.extend([flag] * count)
            flags = flags[:n_points]
            coords = []
            for short_bit, same_bit in ((2, 16), (4, 32)):
                value, axis = 0, []
                for flag in flags:
                    if flag & short_bit:
                        delta = g[at]
                        at += 1
                        value += delta if flag & same_bit else -delta
                    elif not flag & same_bit:
                        value += struct.unpack(">h", g[at:at + 2])[0]
                        at += 2
                    value = (value + 0x8000) % 0x10000 - 0x8000
                    axis.append(value)
                coords.append(axis)
            out, first = [], 0
            for end in ends:
                pts = [(F(coords[0][i]), F(coords[1][i]), bool(flags[i] & 1)) for i in range(first, end + 1)]
                first = end + 1
                out.append((pts, transform))
            return out
        if n_contours < 0:
            out, at = [], 10
            while True:
                flags, component = struct.unpack(">HH", g[at:at + 4])
                at += 4
                a, b, c, d, e, f = F(1), F(0), F(0), F(1), F(0), F(0)
                if flags & 2:
                    if flags & 1:
                        ex, fy = struct.unpack(">hh", g[at:at + 4])
                    else:
                        ex, fy = struct.unpack(">bb", g[at:at + 2])
                    e, f = F(ex), F(fy)
                at += 4 if flags & 1 else 2
                f2 = lambda v: F(v) / F(16384.0)
                if flags & 0x80:
                    va, vb, vc, vd = struct.unpack(">hhhh", g[at:at + 8])
                    a, b, c, d = f2(va), f2(vb), f2(vc), f2(vd)
                    at += 8
                elif flags & 0x40:
                    va, vd = struct.unpack(">hh", g[at:at + 4])
                    a, d = f2(va), f2(vd)
                    at += 4
                elif flags & 8:
                    a = d = f2(struct.unpack(">h", g[at:at + 2])[0])
                    at += 2
                local = (a, b, c, d, e, f)
                combined = local if transform is None else combine(transform, local)
                if is_identity(combined):
                    combined = None
                out.extend(self.contours(component, combined, depth + 1))
                if not flags & 0x20:
                    break
            return out
        return []


def combine(t1, t2):
    a1, b1, c1, d1, e1, f1 = t1
    a2, b2, c2, d2, e2, f2 = t2
    return (F(a1 * a2 + c1 * b2), F(b1 * a2 + d1 * b2), F(a1 * c2 + c1 * d2), F(b1 * c2 + d1 * d2), F(F(a1 * e2 + c1 * f2) + e1), F(F(b1 * e2 + d1 * f2) + f1))


def is_identity(t):
    return t == (F(1), F(0), F(0), F(1), F(0), F(0))


def apply(t, x, y):
    if t is None:
        return x, y
    a, b, c, d, e, f = t
    return F(F(a * x + c * y) + e), F(F(b * x + d * y) + f)


def mid(p, q):
    return (F(p[0] + F(q[0] - p[0]) * F(0.5)), F(p[1] + F(q[1] - p[1]) * F(0.5)))


def contour_to_path(points, transform):
    """One closed TrueType contour -> (start, [(type, floats...)]) in the crate's call order: move_to, line_to / quad_to ..., the
    closing segment, close(). Works on the whole point list: implied on-curve midpoints are made explicit first."""
    n = len(points)
    pending_first_off = None
    if points[0][2]:
        start = points[0][:2]
        rest = list(points[1:])
    elif n > 1 and not points[1][2]:
        start = mid(points[0], points[1])
        pending_first_off = points[0][:2]
        rest = list(points[1:])  # the second off-curve point is the control of the first curve
    else:
        pending_first_off = points[0][:2]
        if n == 1:
            return None  # a lone off-curve point never starts a path: nothing is emitted before close()
        start = points[1][:2]
        rest = list(points[2:])
    segments = []
    control = None
    for x, y, on in rest:
        if control is None:
            if on:
                segments.append(("L", (x, y)))
            else:
                control = (x, y)
        elif on:
            segments.append(("Q", control, (x, y)))
            control = None
        else:
            segments.append(("Q", control, mid(control, (x, y))))
            control = (x, y)
    if pending_first_off is not None and control is not None:
        segments.append(("Q", control, mid(control, pending_first_off)))
        control = None
    if pending_first_off is not None:
        segments.append(("Q", pending_first_off, start))
    elif control is not None:
        segments.append(("Q", control, start))
    else:
        segments.append(("L", start))
    types, floats = [], []
    for seg in segments:
        if seg[0] == "L":
            types.append(0)
            floats.append(tuple(float(v) for v in apply(transform, *seg[1])))
        else:
            types.append(1)
            floats.append(tuple(float(v) for p in seg[1:] for v in apply(transform, *p)))
    return tuple(float(v) for v in apply(transform, *start)), types, floats


def paths_of_glyph(face: Face, gid):
    """-> [(start, types, records)] or [] (text.rs:97-104)."""
    out = []
    lone = False
    for points, transform in face.contours(gid):
        path = contour_to_path(points, transform)
        if path is None:
            lone = True
            out.append(((0.0, 0.0), [], []))  # close() still pushes the (empty) default path
        else:
            out.append(path)
    xs = [v for start, _, recs in out for v in (start[0],) + tuple(r[k] for r in recs for k in range(0, len(r), 2))]
    ys = [v for start, _, recs in out for v in (start[1],) + tuple(r[k] for r in recs for k in range(1, len(r), 2))]
    if not out or (lone and all(len(t) == 0 for _, t, _ in out)):
        return []
    if min(xs) < -32768 or max(xs) > 32767 or min(ys) < -32768 or max(ys) > 32767:
        return []
    return out


def trunc_div(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def aligned_positions(face: Face, size, orientation, major_alignment, minor_alignment, text):
    """calculate_aligned_positions!, text.rs:145-230; enums as integers in declaration order."""
    replacement = face.glyph_index(0xFFFD)
    major_axis, sign_x, sign_y = ((0, -1, -1), (0, 1, -1), (1, 1, -1), (1, 1, 1))[orientation]
    line_minor, gap = (face.height, face.line_gap) if major_axis == 0 else (0, 0)
    lines, line_major, extent, glyphs, prev, index = [], 0, [0, 0], [], None, 0
    for ch in text:
        code = ord(ch) if isinstance(ch, str) else int(ch)
        index += 1
        pos = list(extent)
        pos[major_axis] = line_major
        if code == 10:
            glyphs.append((pos, 0))
            lines.append((index, glyphs))
            glyphs = []
            extent[major_axis] = max(extent[major_axis], line_major)
            extent[1 - major_axis] += line_minor + gap
            line_major, prev = 0, None
        else:
            gid = face.glyph_index(code)
            if gid is None:
                gid = replacement if replacement is not None else 0
            if face.kern is not None and prev is not None and (prev, gid) in face.kern:
                line_major += face.kern[(prev, gid)]
            prev = gid
            advance = face.hor_advance(gid) if major_axis == 0 else None
            if advance is not None:
                line_major += advance
            glyphs.append((pos, gid))
    pos = list(extent)
    pos[major_axis] = line_major
    glyphs.append((pos, 0))
    lines.append((index + 1, glyphs))
    extent[major_axis] = max(extent[major_axis], line_major)
    extent[1 - major_axis] += line_minor
    offset = [0, 0]
    offset[1 - major_axis] = (-face.descender, 0, trunc_div(face.x_height or 0, 2), -line_minor)[minor_alignment]
    for _, line in lines:
        line_extent = line[-1][0][major_axis]
        off = list(offset)
        off[major_axis] = (trunc_div(-extent[major_axis], 2), trunc_div(-line_extent, 2), trunc_div(-line_extent, 2),
                           trunc_div(extent[major_axis], 2) - line_extent)[major_alignment]
        off[1 - major_axis] -= trunc_div(extent[1 - major_axis] - line_minor, 2)
        for p, _ in line:
            p[0] = sign_x * (p[0] + off[0])
            p[1] = sign_y * (p[1] + off[1])
    return extent, [sign_x * offset[0], sign_y * offset[1]], lines


def transform_point(scale, tx, ty, p):
    """Path::transform with a translate2d motor: x' = (tx + x * scale) + y * 0, then SafeFloat canonicalisation of -0."""
    x = F(F(tx + F(p[0]) * scale) + F(p[1]) * F(0.0))
    y = F(F(ty + F(p[0]) * F(0.0)) + F(p[1]) * scale)
    return (float(x) + 0.0, float(y) + 0.0)


def paths_of_text(face: Face, size, orientation, major_alignment, minor_alignment, text):
    """text.rs:236-263 without a clipping area."""
    _, _, lines = aligned_positions(face, size, orientation, major_alignment, minor_alignment, text)
    scale = F(size) / F(face.height)
    out = []
    for _, line in lines:
        for (x, y), gid in line[:-1]:
            tx, ty = F(F(x) * scale), F(F(y) * scale)
            for start, types, records in paths_of_glyph(face, gid):
                new_records = []
                for t, rec in zip(types, records):
                    pts = [transform_point(scale, tx, ty, rec[k:k + 2]) for k in range(0, len(rec), 2)]
                    new_records.append(tuple(v for p in pts for v in p))
                out.append((transform_point(scale, tx, ty, start), list(types), new_records))
    return out
