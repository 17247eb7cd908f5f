"""Glyph producer (SURVEY.md §8 a23 / config 3 input): the native text.rs restatement behind the C ABI (csrc/text.cpp, host code, no GPU
needed) against the independent pure-Python oracle (oracle/text.py) and hand-checkable known answers for the bundled font.
PARITY UNPINNED: the reference has no fixtures for this path; the font itself is the only reference data."""
import os

import numpy as np
import pytest

from contrast_renderer_amd import text as T

FONT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "contrast_renderer_amd", "data", "fonts", "OpenSans-Regular.ttf")


@pytest.fixture(scope="module")
def fonts():
    from oracle import text as O
    data = open(FONT, "rb").read()
    return T.Font("OpenSans", data), O.Face(data), O


def as_tuples(paths):
    return [(tuple(p.start), [int(t) for t in p.segment_types], [tuple(r) for r in p.records]) for p in paths]


def test_metrics_known_answers(fonts):
    font, face, _ = fonts
    # unitsPerEm 2048, hhea ascender 2189 / descender -600 -> height 2789, OS/2 sxHeight 1096, no kern table
    assert (font.units_per_em(), font.ascender(), font.descender(), font.line_gap(), font.height(), font.x_height()) == (2048, 2189, -600, 0, 2789, 1096)
    assert font.number_of_glyphs() == face.n_glyphs == 938
    assert font.vertical_height() is None and font.glyphs_kerning(36, 57) is None
    assert font.name() == "OpenSans" and font.face() is font


def test_cmap_advances_and_boxes_match_the_oracle(fonts):
    font, face, _ = fonts
    codes = list(range(0, 0x250)) + list(range(0x2000, 0x2200)) + [0xFFFD, 0x1F600, 0x10FFFF]
    for code in codes:
        assert font.glyph_index(code) == face.glyph_index(code), hex(code)
    assert font.glyph_index("A") is not None and font.glyph_index(0x1F600) is None
    for gid in range(face.n_glyphs):
        assert font.glyph_hor_advance(gid) == face.hor_advance(gid)
        box = face.bounding_box(gid)
        assert font.glyph_bounding_box(gid) == (tuple(box) if box else None)
    assert font.glyph_hor_advance(938) is None and font.glyph_bounding_box(938) is None


def test_rectangle_glyph_is_four_lines_with_the_closing_line(fonts):
    font, _, _ = fonts
    paths = T.paths_of_glyph(font, font.glyph_index("I"))
    assert len(paths) == 1
    p = paths[0]
    # a TrueType contour does not repeat its first point: the walk closes it with an explicit line_to(start) before close()
    assert p.start == (201.0, 0.0) and [int(t) for t in p.segment_types] == [0, 0, 0, 0]
    assert p.records == [(201.0, 1462.0), (371.0, 1462.0), (371.0, 0.0), (201.0, 0.0)]
    assert p.stroke_options is None
    assert T.paths_of_glyph(font, font.glyph_index(" ")) == []  # no outline -> Vec::new()


def test_every_glyph_outline_matches_the_oracle(fonts):
    font, face, O = fonts
    n_paths = n_composite = 0
    for gid in range(face.n_glyphs):
        want = O.paths_of_glyph(face, gid)
        got = as_tuples(T.paths_of_glyph(font, gid))
        assert got == [(tuple(s), t, [tuple(r) for r in recs]) for s, t, recs in want], f"glyph {gid}"
        n_paths += len(got)
        data = face.glyph_data(gid)
        n_composite += bool(data) and int.from_bytes(data[0:2], "big", signed=True) < 0
    assert n_paths > 1200 and n_composite > 100  # simple and composite glyphs are both exercised
    # every contour is closed: the last control point of a path is its start
    for gid in (font.glyph_index(c) for c in "aB8&@é"):
        for p in T.paths_of_glyph(font, gid):
            assert p.records[-1][-2:] == p.start


@pytest.mark.parametrize("orientation", list(T.Orientation))
def test_aligned_positions_match_the_oracle(fonts, orientation):
    font, face, O = fonts
    text = "Hello, MI355X!\nsecond line\n\nyo �世"
    for major in T.Alignment:
        for minor in T.Alignment:
            layout = T.Layout(24.0, orientation, major, minor)
            extent, offset, lines = T.calculate_aligned_positions(font, layout, text)
            oextent, ooffset, olines = O.aligned_positions(face, 24.0, int(orientation), int(major), int(minor), text)
            assert extent == oextent and offset == ooffset
            assert lines == [(end, [((p[0], p[1]), g) for p, g in line]) for end, line in olines]
    # structure: one entry per character plus one terminator per line; line ends count characters
    assert sum(len(line) for _, line in lines) == len(text) + 1 and [end for end, _ in lines] == [15, 27, 28, len(text) + 1]


def test_paths_of_text_matches_the_oracle_and_the_hand_computed_placement(fonts):
    font, face, O = fonts
    text = "Hi\nyo."
    for size in (12.0, 31.5):
        for major, minor in ((T.Alignment.Begin, T.Alignment.Baseline), (T.Alignment.Center, T.Alignment.Center), (T.Alignment.End, T.Alignment.Begin)):
            layout = T.Layout(size, T.Orientation.LeftToRight, major, minor)
            got = as_tuples(T.paths_of_text(font, layout, text))
            want = O.paths_of_text(face, size, 1, int(major), int(minor), text)
            assert got == [(tuple(s), t, [tuple(r) for r in recs]) for s, t, recs in want]
    # hand check: size == height -> scale 1; single glyph, Baseline/Begin: origin x = -advance/2 (trunc), y = 0
    layout = T.Layout(2789.0, T.Orientation.LeftToRight, T.Alignment.Begin, T.Alignment.Baseline)
    p = T.paths_of_text(font, layout, "I")[0]
    advance = font.glyph_hor_advance(font.glyph_index("I"))
    assert p.start == (201.0 - advance // 2, 0.0 - 0.0) or p.start == (201.0 - advance // 2, 0.0)


def test_clipping_area_discards_glyphs_outside(fonts):
    font, _, _ = fonts
    layout = T.Layout(32.0, T.Orientation.LeftToRight, T.Alignment.Begin, T.Alignment.Baseline)
    text = "IIIIIIII"
    every = as_tuples(T.paths_of_text(font, layout, text))
    assert len(every) == 8
    # clockwise (y-up) polygon covering x < 0 only (utils.rs:83-98 expects clockwise vertices)
    left = [(-1000.0, -1000.0), (-1000.0, 1000.0), (0.0, 1000.0), (0.0, -1000.0)]
    kept = as_tuples(T.paths_of_text(font, layout, text, clipping_area=left))
    assert 0 < len(kept) < 8 and all(k in every for k in kept)
    assert all(k[0][0] < 0.0 for k in kept)  # glyphs that start right of the area are gone; the others keep their position
    far = [(5000.0, 5000.0), (5000.0, 6000.0), (6000.0, 6000.0), (6000.0, 5000.0)]
    assert T.paths_of_text(font, layout, text, clipping_area=far) == []


def test_path_list_transform(fonts):
    font, _, _ = fonts
    gid = font.glyph_index("o")
    base = as_tuples(T.paths_of_glyph(font, gid))
    f = np.float32
    v = (f(3.25), f(-7.5))
    moved = as_tuples(T.glyph_path_list(font, gid).transform(0.5, (1.0, 0.0, -0.5 * float(v[1]), 0.5 * float(v[0]))).to_paths())  # translate2d, utils.rs:127-129
    for (s0, t0, r0), (s1, t1, r1) in zip(base, moved):
        assert t0 == t1
        assert s1 == (float(v[0] + f(s0[0]) * f(0.5)), float(v[1] + f(s0[1]) * f(0.5)))
        for a, b in zip(r0, r1):
            assert b == tuple(float((v[k % 2] + f(a[k]) * f(0.5))) for k in range(len(a)))
    # a rotor keeps distances from the origin (rotate2d, utils.rs:122-125)
    angle = 0.7
    turned = as_tuples(T.glyph_path_list(font, gid).transform(1.0, (np.cos(angle / 2), np.sin(angle / 2), 0.0, 0.0)).to_paths())
    for (s0, _, _), (s1, _, _) in zip(base, turned):
        assert abs(np.hypot(*s0) - np.hypot(*s1)) < 1e-2 and s0 != s1


def test_malformed_fonts_are_rejected_not_crashed(fonts):
    from contrast_renderer_amd import ContrastError
    data = open(FONT, "rb").read()
    for bad in (b"", b"not a font at all", data[:40], b"\x00\x01\x00\x00" + b"\xff" * 64):
        with pytest.raises(ContrastError):
            T.Font("bad", bad)
    # truncated / garbled glyph data must not read out of bounds: halve the glyf table's length in the directory, then scramble it
    at = data.index(b"glyf", 12, 12 + 16 * 32)
    length = int.from_bytes(data[at + 12:at + 16], "big")
    offset = int.from_bytes(data[at + 8:at + 12], "big")
    cut = T.Font("cut", data[:at + 12] + (length // 2).to_bytes(4, "big") + data[at + 16:])
    rng = np.random.RandomState(1)
    noise = bytearray(data)
    noise[offset:offset + length] = rng.randint(0, 256, length, dtype=np.uint8).tobytes()
    noisy = T.Font("noise", bytes(noise))
    for gid in range(cut.number_of_glyphs()):
        T.paths_of_glyph(cut, gid)
        T.paths_of_glyph(noisy, gid)


def test_text_geometry_and_its_cursor_helpers(fonts):
    """TextGeometry (text.rs:266-347): every line carries one more glyph position than printable characters (its line break), lines are
    one `size` apart along the minor axis, and the three cursor helpers follow the reference's definitions."""
    native = fonts[0]
    layout = T.Layout(1.0, T.Orientation.LeftToRight, T.Alignment.Center, T.Alignment.Center)
    g = T.TextGeometry.new(native, layout, "ab\ncd\nef")
    assert g.major_axis == 0 and g.half_extent[1] == 1.5 and [end for end, _ in g.lines] == [3, 6, 9]
    assert [len(p) for _, p in g.lines] == [3, 3, 3] and [p[0][1] for _, p in g.lines] == [1.0, 0.0, -1.0]
    for _, positions in g.lines:  # centred (integer font units: symmetric up to one unit)
        assert abs(positions[0][0] + positions[-1][0]) < 1e-3 and positions[0][0] < positions[1][0] < positions[2][0]
    assert [g.line_index_from_char_index(i) for i in range(9)] == [0, 0, 0, 1, 1, 1, 2, 2, 2]
    with pytest.raises(IndexError):
        g.line_index_from_char_index(9)
    # a cursor exactly on a glyph position selects that character; halfway past it the next one
    for line, (end, positions) in enumerate(g.lines):
        first = 0 if line == 0 else g.lines[line - 1][0]
        for k, p in enumerate(positions):
            assert g.char_index_from_position(p) == first + k
        assert g.char_index_from_position((positions[0][0] * 0.49 + positions[1][0] * 0.51, positions[0][1])) == first + 1
    assert g.char_index_from_position((-5.0, 9.0)) == 0 and g.char_index_from_position((5.0, -9.0)) == 8
    # up / down keep the horizontal position; the first and last line clamp to the text's ends
    assert g.advance_char_index_by_line_index(4, -1) == 1 and g.advance_char_index_by_line_index(4, 1) == 7
    assert g.advance_char_index_by_line_index(1, -1) == 0 and g.advance_char_index_by_line_index(7, 1) == 8
    assert g.advance_char_index_by_line_index(0, 2) == 6
    vertical = T.TextGeometry.new(native, T.Layout(2.0, T.Orientation.TopToBottom, T.Alignment.Begin, T.Alignment.Begin), "ab\nc")
    assert vertical.major_axis == 1 and [end for end, _ in vertical.lines] == [3, 5]
    assert T.byte_offset_of_char_index("aé€b", 2) == 3 and T.byte_offset_of_char_index("aé€b", 3) == 6
    assert T.byte_offset_of_char_index("aé€b", 10) == 7
