"""VERDICT r03 item 8 asked for a lane-parallel form of convex_hull::andrew's chain walk (convex_hull.rs:7-40): "pop rounds" — every interior
point of a chain tested against its CURRENT neighbours with the same `<= ERROR_MARGIN` predicate, round after round until nothing changes
— "checked against andrew; if no exact formulation exists, say so with the counter-example committed as a test". This is that test.

No exact formulation of that kind exists: the predicate (a v b) v c is evaluated in f32 on un-translated coordinates, so for a glyph set
at x = -975, y = 1017 its rounding noise is ~0.1 — a thousand times ERROR_MARGIN — and WHICH triples are evaluated decides which of
several locally convex chains comes out. The serial walk tests a point against the two points on top of the stack at that moment; a pop
round tests it against its neighbours in the sorted order. On scenes.scene_glyphs(50 000) the three round formulations tried leave another
hull than the walk on 14 515 / 4 012 / 1 482 Shapes (tools/proto_hull.py prints the counts for four scenes). The fixture is a small one of
them: ten candidates on which all three differ. The hull kernels therefore keep the serial walk (csrc/tessellate.hip, a lane pair per Shape)."""
import json
import os

import numpy as np

f = np.float32
ERROR_MARGIN = f(1e-4)


def turn(a, b, c):
    """regressive(regressive(vec_to_point(a), vec_to_point(b)), vec_to_point(c)) with Point = (1, x, y): oracle/ga.hpp:55-59, every
    operation rounded to f32, sums left to right (the reference's expression, convex_hull.rs:17)."""
    one = f(1.0)
    l0 = f(f(a[1] * b[0]) - f(a[0] * b[1]))
    l1 = f(f(one * b[1]) - f(a[1] * one))
    l2 = f(f(a[0] * one) - f(one * b[0]))
    return f(f(f(l0 * one) + f(l1 * c[0])) + f(l2 * c[1]))


def walk(p, order):
    """one chain of Andrew's scan: the serial walk"""
    stack = []
    for i in order:
        while len(stack) > 1 and turn(p[stack[-2]], p[stack[-1]], p[i]) <= ERROR_MARGIN:
            stack.pop()
        stack.append(i)
    return stack


def pop_rounds(p, order, mode):
    """mode 0: every removable interior point leaves at once; 1 / 2: of two removable neighbours only the later / the earlier one"""
    alive = list(order)
    while len(alive) >= 3:
        n = len(alive)
        leaves = [False] + [bool(turn(p[alive[k - 1]], p[alive[k]], p[alive[k + 1]]) <= ERROR_MARGIN) for k in range(1, n - 1)] + [False]
        if not any(leaves):
            break
        if mode == 1:
            go = [leaves[k] and not leaves[k + 1] for k in range(n - 1)] + [False]
        elif mode == 2:
            go = [False] + [leaves[k] and not leaves[k - 1] for k in range(1, n)]
        else:
            go = leaves
        alive = [a for a, g in zip(alive, go) if not g]
    return alive


def chains(candidates):
    """-> (the hull of the serial walk, [the hulls of the three round formulations]) as lists of (x, y): lower chain without its last
    point, then the upper chain without its last point (convex_hull.rs:23, :34)"""
    c = np.asarray(candidates, dtype=np.float32)
    p = c[sorted(range(len(c)), key=lambda i: (c[i][0], c[i][1]))]  # SafeFloat's lexicographic order (safe_float.rs:163-173)
    up = list(range(len(p)))
    down = up[::-1]

    def hull(chain_of):
        return [tuple(map(float, p[i])) for i in chain_of(up)[:-1] + chain_of(down)[:-1]]
    return hull(lambda order: walk(p, order)), [hull(lambda order, m=mode: pop_rounds(p, order, m)) for mode in (0, 1, 2)]


def fixture():
    with open(os.path.join(os.path.dirname(__file__), "golden", "hull_pop_rounds_counterexample.json")) as fh:
        bits = np.array(json.load(fh)["candidates_f32_bits"], dtype=np.uint32)
    return bits.view(np.float32)


def test_the_serial_walk_of_this_file_is_the_oracles_andrew():
    """A closed polygon through the fixture's points has exactly them (and its start point once more) as hull candidates
    (fill.rs: the start point and every line's end point): the oracle's hull of it equals the walk restated above."""
    from contrast_renderer_amd import Path, batch_from_shapes
    from oracle import Oracle
    c = fixture()
    path = Path(start=tuple(map(float, c[0])))
    for x, y in c[1:]:
        path.push_line((float(x), float(y)))
    path.push_line(tuple(map(float, c[0])))
    o = Oracle(batch_from_shapes([([], [path])]))
    assert o.status() == 0
    vo, _, vb, _ = o.shape(0)
    strip = vb[int(vo[6]):int(vo[7])].view(np.float32).reshape(-1, 2)
    h = len(strip)  # triangle_fan_to_strip (vertex.rs:28-35): [0, h-1, 1, h-2, ...] — undone here
    fan = [None] * h
    for i in range(h):
        fan[i // 2 if i % 2 == 0 else h - 1 - i // 2] = tuple(map(float, strip[i]))
    serial, _ = chains(np.concatenate([c, c[:1]]))
    assert fan == serial


def test_pop_rounds_leave_another_hull_than_the_serial_walk():
    serial, by_rounds = chains(fixture())
    for mode, rounds in enumerate(by_rounds):
        assert rounds != serial, f"pop rounds (mode {mode}) reproduce the serial walk on the fixture"
        assert rounds[0] == serial[0]  # (the same first vertex: they differ in which interior candidates survive)
