// tools/proto_edges.cpp — DEVELOPMENT TOOL (CPU): the boundary-edge + backdrop formulation of the solid fill and of the hull cover,
// written the way csrc/raster_edges.hip evaluates it (tile-relative f32 arithmetic, per-tile backdrop at q0, path q0 -> q_k -> p), checked
// bit for bit against the triangle-strip specification of oracle/raster.hpp. Not shipped, not used by tests; it exists so that the sign
// conventions and tie rules can be verified without a GPU. Build + run: tools/proto_edges.py
// Mode 2 (round 4) states the same sum the way the row-span raster kernel accumulates it: per (edge, sample row) ONE switch column found by
// bisection of the exact predicate (g is a step function of x along a sample row: E = fma(ry, bx, fma(rx, nay, c)) is monotone in rx under
// round-to-nearest), two deposits into a 16 x 16 delta grid per tile — the row constant at column 0, +-1 at the switch column — and a prefix
// sum along every row. It must equal mode 1 (and mode 0, the oracle's strips) sample for sample.
#include <cstdio>

#include "../oracle/api.cpp"

namespace proto {
using namespace oracle;

struct Edge { // canonical endpoints lo < hi (lexicographic), directed flag
    float lo[2], hi[2], bx, nay;
    int sigma; // -1: the chain runs lo -> hi, +1: hi -> lo
    bool tl;   // top-left flag of the canonical direction
    bool down; // canonical dy > 0
};

static bool make_edge(const float a[2], const float b[2], Edge& e) { // the chain runs a -> b
    if (a[0] == b[0] && a[1] == b[1]) return false;
    if (!(a[0] == a[0] && a[1] == a[1] && b[0] == b[0] && b[1] == b[1])) return false;
    const bool flip = !lex_less(a, b);
    const float* lo = flip ? b : a;
    const float* hi = flip ? a : b;
    e.lo[0] = lo[0], e.lo[1] = lo[1], e.hi[0] = hi[0], e.hi[1] = hi[1];
    e.bx = hi[0] - lo[0];
    e.nay = -(hi[1] - lo[1]);
    e.sigma = flip ? 1 : -1;
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1];
    e.tl = dy < 0.0f || (dy == 0.0f && dx > 0.0f);
    e.down = dy > 0.0f;
    return true;
}
// chain of a strip with n positions: see DESIGN.md (boundary of the zig-zag strip)
static void strip_chain(const std::vector<const float*>& pos, std::vector<Edge>& out) {
    const size_t n = pos.size();
    if (n < 3) return;
    for (size_t i = 0; i < n; ++i) {
        size_t j;
        if (i == 0) j = 1;
        else if ((i & 1) == 0) j = i - 2;
        else if (i + 2 <= n - 1) j = i + 2;
        else j = (i == n - 1) ? n - 2 : n - 1;
        Edge e;
        if (make_edge(pos[i], pos[j], e)) out.push_back(e);
    }
}
static inline bool g_of(float E, bool tl) { return E > 0.0f || (E == 0.0f && tl); }

struct TileEval {
    float tx0, ty0;
    float c;
    const Edge* e;
    void set(const Edge& ed, int tx, int ty) {
        e = &ed;
        tx0 = (float)(tx * TILE), ty0 = (float)(ty * TILE);
        c = ed.bx * (ty0 - ed.lo[1]) + ed.nay * (tx0 - ed.lo[0]);
    }
    bool g(float rx, float ry) const { return g_of(fmaf(ry, e->bx, fmaf(rx, e->nay, c)), e->tl); }
};

// winding of the chain `edges` at every sample of the frame, accumulated into acc[y][x][s] (int), the GPU way
static bool g_rows = false; // mode 2: the row-span accumulation
// lower bound of a monotone predicate over the 16 columns of a sample row: 5 exact evaluations (17 possible answers)
template <class P>
static int first_true(P pred) {
    int lo = 0;
    for (int step = 8; step >= 1; step >>= 1)
        if (!pred(lo + step - 1)) lo += step;
    if (lo == 15 && !pred(15)) lo = 16;
    return lo;
}
static void chain_winding(const Frame& f, const std::vector<Edge>& edges, std::vector<int>& acc, int& tiles_touched, long& pairs) {
    if (edges.empty()) return;
    const int W = (int)f.width, H = (int)f.height;
    float minx = INFINITY, maxx = -INFINITY, miny = INFINITY, maxy = -INFINITY;
    for (const Edge& e : edges) {
        minx = std::fmin(minx, e.lo[0]), maxx = std::fmax(maxx, e.hi[0]);
        miny = std::fmin(miny, std::fmin(e.lo[1], e.hi[1])), maxy = std::fmax(maxy, std::fmax(e.lo[1], e.hi[1]));
    }
    const int x0 = (int)std::floor(std::fmin(std::fmax(minx, 0.0f), (float)W)), x1 = (int)std::floor(std::fmax(std::fmin(maxx, (float)(W - 1)), -1.0f));
    const int y0 = (int)std::floor(std::fmin(std::fmax(miny, 0.0f), (float)H)), y1 = (int)std::floor(std::fmax(std::fmin(maxy, (float)(H - 1)), -1.0f));
    if (x0 > x1 || y0 > y1) return;
    const uint32_t S = f.samples;
    float sox[4], soy[4];
    for (uint32_t s = 0; s < S; ++s) sample_offset(S, s, sox[s], soy[s]);
    float ry_first = soy[0], ry_last = 15.0f + soy[S - 1], rx_last = 0.0f;
    for (uint32_t s = 0; s < S; ++s) rx_last = std::fmax(rx_last, 15.0f + sox[s]);
    for (int ty = y0 / TILE; ty <= y1 / TILE; ++ty)
        for (int tx = x0 / TILE; tx <= x1 / TILE; ++tx) {
            ++tiles_touched;
            const float tx0 = (float)(tx * TILE), ty0 = (float)(ty * TILE);
            // ---- backdrop: the ray formula at q0 = (tx0, ty0 + ry_first), every edge
            int bd = 0;
            std::vector<const Edge*> touching;
            for (const Edge& e : edges) {
                TileEval t;
                t.set(e, tx, ty);
                const float ymin = std::fmin(e.lo[1], e.hi[1]), ymax = std::fmax(e.lo[1], e.hi[1]);
                const float q0y = ty0 + ry_first;
                const bool Y0 = ymin <= q0y && q0y < ymax;
                if (Y0) bd += e.sigma * ((t.g(0.0f, ry_first) ? 1 : 0) - (e.down ? 1 : 0));
                // ---- does the edge matter inside the tile? g not constant over the tile's evaluation points and the boxes overlap
                const float rx_hi = e.nay > 0.0f ? rx_last : 0.0f, rx_lo = e.nay > 0.0f ? 0.0f : rx_last;
                const bool gmax = t.g(rx_hi, ry_last), gmin = t.g(rx_lo, ry_first); // bx >= 0: E grows with ry
                const bool box = e.lo[0] <= tx0 + rx_last && e.hi[0] >= tx0 && ymin <= ty0 + ry_last && ymax >= ty0 + ry_first;
                if (gmax != gmin && box) touching.push_back(&e);
            }
            pairs += (long)touching.size();
            if (g_rows) {
                // ---- the row-span form. Sample rows of the tile: pixel row r, sample s -> row index r * S + s; the columns of a row are the 16
                //      pixels at that sample's x offset. Per (edge, row): d(j) = sigma * [Y_k g(j) + A_k] with the row constant
                //      A_k = xr (g(q_k) - g(q_0)) - Y_k g(q_k); along the row g(j) = [j >= s] (nay >= 0) or [j < s] (nay < 0).
                std::vector<int> grid((size_t)TILE * S * TILE, 0); // delta form: w(row, j) = bd + sum_{i <= j} grid[row][i]
                for (const Edge* ep : touching) {
                    const Edge& e = *ep;
                    TileEval t;
                    t.set(e, tx, ty);
                    const float ymin = std::fmin(e.lo[1], e.hi[1]), ymax = std::fmax(e.lo[1], e.hi[1]);
                    const bool xr = e.lo[0] <= tx0 && tx0 < e.hi[0];
                    const int gq0 = t.g(0.0f, ry_first);
                    for (int r = 0; r < TILE; ++r)
                        for (uint32_t sm = 0; sm < S; ++sm) {
                            const float ry = (float)r + soy[sm], sy = ty0 + ry;
                            const bool Yk = ymin <= sy && sy < ymax;
                            const int gqk = t.g(0.0f, ry);
                            const int A = (xr ? gqk - gq0 : 0) - (Yk ? gqk : 0);
                            int* row = &grid[((size_t)r * S + sm) * TILE];
                            if (!Yk) {
                                row[0] += e.sigma * A;
                                continue;
                            }
                            const bool inv = e.nay < 0.0f; // g falls along the row
                            const int sw = first_true([&](int j) { return t.g((float)j + sox[sm], ry) != inv; });
                            const int g_left = (sw == 0) != inv ? 1 : 0; // g at column 0
                            row[0] += e.sigma * (A + g_left);
                            if (sw > 0 && sw < TILE) row[sw] += e.sigma * (inv ? -1 : 1);
                        }
                }
                for (int py = std::max(0, ty * TILE); py < std::min(H, ty * TILE + TILE); ++py)
                    for (uint32_t sm = 0; sm < S; ++sm) {
                        const int* row = &grid[((size_t)(py - ty * TILE) * S + sm) * TILE];
                        int run = bd;
                        for (int j = 0; j < TILE; ++j) {
                            run += row[j];
                            const int px = tx * TILE + j;
                            if (px >= 0 && px < W) acc[((size_t)py * f.width + px) * S + sm] += run;
                        }
                    }
                continue;
            }
            // ---- per sample: w = bd + sum over touching edges of sigma * [xr (g(qk) - g(q0)) + Yk (g(p) - g(qk))]
            for (int py = std::max(0, ty * TILE); py < std::min(H, ty * TILE + TILE); ++py)
                for (int px = std::max(0, tx * TILE); px < std::min(W, tx * TILE + TILE); ++px)
                    for (uint32_t s = 0; s < S; ++s) {
                        const float rx = (float)(px - tx * TILE) + sox[s], ry = (float)(py - ty * TILE) + soy[s];
                        int w = bd;
                        for (const Edge* ep : touching) {
                            const Edge& e = *ep;
                            TileEval t;
                            t.set(e, tx, ty);
                            const float ymin = std::fmin(e.lo[1], e.hi[1]), ymax = std::fmax(e.lo[1], e.hi[1]);
                            const float sy = ty0 + ry;
                            const bool Yk = ymin <= sy && sy < ymax;
                            const bool xr = e.lo[0] <= tx0 && tx0 < e.hi[0];
                            const int gq0 = t.g(0.0f, ry_first), gqk = t.g(0.0f, ry), gp = t.g(rx, ry);
                            w += e.sigma * ((xr ? gqk - gq0 : 0) + (Yk ? gp - gqk : 0));
                        }
                        acc[((size_t)py * f.width + px) * S + s] += w;
                    }
        }
}

static void transform_points(const Frame& f, const float m[16], const std::vector<Vertex0>& v, std::vector<float>& out) {
    out.resize(v.size() * 2);
    for (size_t i = 0; i < v.size(); ++i) to_framebuffer(m, (float)f.width, (float)f.height, v[i].p, &out[2 * i]);
}

// Shape::render(Stencil) with the solid strips replaced by their boundary chain
static void render_stencil_edges(Frame& f, const Shape& shape, const float m[16], int& tiles, long& pairs) {
    // strokes and curves: the oracle's own code, on a copy of the shape without solid strips
    Shape rest = shape;
    rest.fill.solid_vertices.clear();
    rest.fill.solid_indices.clear();
    rest.fill.solid_restarts.clear();
    render_stencil(f, rest, m);
    // NOTE: order differs from the reference (solid after curves) — integer adds commute, and strokes stay first
    std::vector<float> pts;
    transform_points(f, m, shape.fill.solid_vertices, pts);
    const auto& idx = shape.fill.solid_indices;
    const auto& restarts = shape.fill.solid_restarts;
    std::vector<Edge> edges;
    size_t run_start = 0, vertex_base = 0, next_restart = 0;
    for (size_t k = 0; k <= idx.size(); ++k) {
        if (k == idx.size() || (next_restart < restarts.size() && restarts[next_restart] == k)) {
            std::vector<const float*> pos;
            for (size_t i = 0; run_start + i < k; ++i) pos.push_back(&pts[2 * (vertex_base + i)]);
            strip_chain(pos, edges);
            vertex_base += k - run_start;
            run_start = k + 1;
            ++next_restart;
        }
    }
    std::vector<int> acc((size_t)f.width * f.height * f.samples, 0);
    chain_winding(f, edges, acc, tiles, pairs);
    for (size_t i = 0; i < acc.size(); ++i)
        if (acc[i]) f.winding[i] = wrap_add(f.winding[i], acc[i], f.winding_mask);
}
static void render_color_edges(Frame& f, const Shape& shape, const float m[16], const float rgba[4], int& tiles, long& pairs) {
    const float src[4] = {rgba[0] * rgba[3], rgba[1] * rgba[3], rgba[2] * rgba[3], rgba[3]};
    const float one_minus_a = 1.0f - src[3];
    std::vector<float> pts;
    transform_points(f, m, shape.convex_hull, pts);
    std::vector<const float*> pos;
    for (size_t i = 0; i < shape.convex_hull.size(); ++i) pos.push_back(&pts[2 * i]);
    std::vector<Edge> edges;
    strip_chain(pos, edges);
    std::vector<int> acc((size_t)f.width * f.height * f.samples, 0);
    chain_winding(f, edges, acc, tiles, pairs);
    for (size_t si = 0; si < acc.size(); ++si) {
        if (!acc[si]) continue;
        const uint32_t st = f.winding[si];
        float* dst = &f.color[si * 4];
        if ((st & f.winding_mask) != 0)
            for (int c = 0; c < 4; ++c) dst[c] = src[c] + dst[c] * one_minus_a;
        f.winding[si] = (uint8_t)(st & ~f.winding_mask);
    }
}
} // namespace proto

extern "C" {
// mode 0: oracle (triangle strips), 1: edge formulation per sample, 2: edge formulation as row spans (switch columns + row prefix sums). Outputs RGBA8 and the final stencil bytes [h][w][msaa].
int proto_render(void* h, uint32_t width, uint32_t height, uint32_t msaa, uint32_t winding_bits, const float* transforms, const float* colors,
                 uint32_t shape_begin, uint32_t shape_end, int mode, uint8_t* rgba8, uint8_t* winding_out, long* stats) {
    Scene* sc = static_cast<Scene*>(h);
    Frame f;
    f.create(width, height, msaa, winding_bits);
    int tiles = 0;
    long pairs = 0;
    for (uint32_t s = shape_begin; s < shape_end && s < sc->shapes.size(); ++s) {
        proto::g_rows = mode == 2;
        if (mode == 0) {
            render_stencil(f, sc->shapes[s], transforms + 16 * (size_t)s);
            render_color(f, sc->shapes[s], transforms + 16 * (size_t)s, colors + 4 * (size_t)s);
        } else {
            proto::render_stencil_edges(f, sc->shapes[s], transforms + 16 * (size_t)s, tiles, pairs);
            proto::render_color_edges(f, sc->shapes[s], transforms + 16 * (size_t)s, colors + 4 * (size_t)s, tiles, pairs);
        }
    }
    resolve_rgba8(f, rgba8);
    if (winding_out) std::copy(f.winding.begin(), f.winding.end(), winding_out);
    if (stats) stats[0] = tiles, stats[1] = pairs;
    return 0;
}
}

// debug: for shape s and pixel (px, py), sample 0 of msaa 1: every solid strip triangle that contains the sample, and every chain edge
// with its terms
extern "C" void proto_probe(void* h, uint32_t width, uint32_t height, const float* transforms, uint32_t s, int px, int py, int hull) {
    using namespace oracle;
    using namespace proto;
    Scene* sc = static_cast<Scene*>(h);
    const Shape& shape = sc->shapes[s];
    const float* m = transforms + 16 * (size_t)s;
    Frame f;
    f.create(width, height, 1, 4);
    std::vector<float> pts;
    transform_points(f, m, hull ? shape.convex_hull : shape.fill.solid_vertices, pts);
    std::vector<uint16_t> hull_idx(shape.convex_hull.size(), 0);
    std::vector<uint32_t> hull_restarts;
    const auto& idx = hull ? hull_idx : shape.fill.solid_indices;
    const auto& restarts = hull ? hull_restarts : shape.fill.solid_restarts;
    const int tx = px / TILE, ty = py / TILE;
    const float rx = (float)(px - tx * TILE) + 0.5f, ry = (float)(py - ty * TILE) + 0.5f;
    size_t run_start = 0, vertex_base = 0, next_restart = 0;
    int strip = 0;
    for (size_t k = 0; k <= idx.size(); ++k) {
        if (k == idx.size() || (next_restart < restarts.size() && restarts[next_restart] == k)) {
            const size_t n = k - run_start;
            printf("strip %d: %zu vertices\n", strip, n);
            for (size_t i = 0; i + 2 < n; ++i) {
                size_t tri[3];
                StripWalker::triangle(i, tri);
                float v[3][2];
                for (int c = 0; c < 3; ++c) v[c][0] = pts[2 * (vertex_base + tri[c])], v[c][1] = pts[2 * (vertex_base + tri[c]) + 1];
                const TriangleSetup t = setup_triangle(v, (int)width, (int)height);
                const float d1x = v[1][0] - v[0][0], d1y = v[1][1] - v[0][1], d2x = v[2][0] - v[0][0], d2y = v[2][1] - v[0][1];
                const float det = d1x * d2y - d2x * d1y;
                if (!t.valid) {
                    printf("  tri %zu INVALID det %.9g  (%.9g,%.9g) (%.9g,%.9g) (%.9g,%.9g)\n", i, det, v[0][0], v[0][1], v[1][0], v[1][1], v[2][0], v[2][1]);
                    continue;
                }
                if (px < t.x0 || px > t.x1 || py < t.y0 || py > t.y1) continue;
                const float tx0 = (float)(tx * TILE), ty0 = (float)(ty * TILE);
                bool inside = true;
                float ev[3];
                for (int e = 0; e < 3; ++e) {
                    const float c = t.e[e].bx * (ty0 - t.e[e].lo[1]) + t.e[e].nay * (tx0 - t.e[e].lo[0]);
                    float E = fmaf(ry, t.e[e].bx, fmaf(rx, t.e[e].nay, c));
                    if (t.e[e].flip) E = -E;
                    ev[e] = E;
                    inside = inside && (E > 0.0f || (E == 0.0f && t.e[e].topleft));
                }
                printf("  tri %zu det %.9g front %d inside %d  E = %.9g %.9g %.9g\n", i, det, (int)t.front, (int)inside, ev[0], ev[1], ev[2]);
            }
            std::vector<const float*> pos;
            for (size_t i = 0; i < n; ++i) pos.push_back(&pts[2 * (vertex_base + i)]);
            std::vector<Edge> edges;
            strip_chain(pos, edges);
            int w = 0;
            for (const Edge& e : edges) {
                TileEval t;
                t.set(e, tx, ty);
                const float ymin = std::fmin(e.lo[1], e.hi[1]), ymax = std::fmax(e.lo[1], e.hi[1]);
                const float sy = (float)(ty * TILE) + ry;
                const bool Y = ymin <= sy && sy < ymax;
                const float E = fmaf(ry, e.bx, fmaf(rx, e.nay, t.c));
                const int term = Y ? e.sigma * ((g_of(E, e.tl) ? 1 : 0) - (e.down ? 1 : 0)) : 0;
                w += term;
                if (Y) printf("  edge (%.9g,%.9g)-(%.9g,%.9g) sigma %d E %.9g term %d\n", e.lo[0], e.lo[1], e.hi[0], e.hi[1], e.sigma, E, term);
            }
            printf("  ray winding of this strip: %d\n", w);
            vertex_base += n;
            run_start = k + 1;
            ++next_restart;
            ++strip;
        }
    }
}
