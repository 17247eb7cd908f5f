// DEVELOPMENT TOOL (CPU). VERDICT r03 item 8 asks whether the monotone-chain walk of convex_hull::andrew (convex_hull.rs:7-40) — inherently
// serial: every pop is decided against the two points that happen to be on top of the stack at that moment — has an exact lane-parallel
// form: "pop rounds" in which every interior point of a chain is tested against its current neighbours with the same `<= ERROR_MARGIN`
// predicate, round after round until nothing changes. This restates both on the oracle's candidates (the same f32 expression, the same
// operand order) and compares the hulls vertex for vertex. Built and driven by tools/proto_hull.py.
#include "../oracle/api.cpp"

namespace proto {
using namespace oracle;

inline bool candidate_less(const Safe2& a, const Safe2& b) { return a.v[0] != b.v[0] ? a.v[0] < b.v[0] : a.v[1] < b.v[1]; }
inline float turn(const Safe2& a, const Safe2& b, const Safe2& c) { return regressive(regressive(vec_to_point(a.v), vec_to_point(b.v)), vec_to_point(c.v)); }

// one chain of Andrew's scan over the sorted points taken in `order` (ascending for the lower chain, descending for the upper one): the
// serial walk (convex_hull.rs:16-22 / :24-33 with the stack floor at the chain's own first point)
inline std::vector<uint32_t> chain_serial(const std::vector<Safe2>& pts, const std::vector<uint32_t>& order) {
    std::vector<uint32_t> st;
    for (uint32_t i : order) {
        while (st.size() > 1 && turn(pts[st[st.size() - 2]], pts[st[st.size() - 1]], pts[i]) <= ERROR_MARGIN) st.pop_back();
        st.push_back(i);
    }
    return st;
}
// Pop rounds. mode 0: every interior point whose turn against its current neighbours is <= ERROR_MARGIN leaves at once.
// mode 1: the same, but of two neighbours that would both leave only the one later in walk order leaves (Andrew pops the point under the
// newcomer: in a run of removable points the last one goes first) — never two adjacent points in one round.
// mode 2: as mode 1 with the earlier one leaving.
inline std::vector<uint32_t> chain_rounds(const std::vector<Safe2>& pts, const std::vector<uint32_t>& order, int mode, uint32_t* rounds) {
    std::vector<uint32_t> alive = order;
    for (;;) {
        const size_t n = alive.size();
        if (n < 3) break;
        std::vector<char> leaves(n, 0);
        bool any = false;
        for (size_t k = 1; k + 1 < n; ++k) leaves[k] = turn(pts[alive[k - 1]], pts[alive[k]], pts[alive[k + 1]]) <= ERROR_MARGIN, any = any || leaves[k];
        if (!any) break;
        *rounds += 1;
        std::vector<char> go = leaves;
        if (mode == 1)
            for (size_t k = 1; k + 1 < n; ++k) go[k] = leaves[k] && !leaves[k + 1];
        if (mode == 2)
            for (size_t k = 1; k + 1 < n; ++k) go[k] = leaves[k] && !leaves[k - 1];
        std::vector<uint32_t> next;
        for (size_t k = 0; k < n; ++k)
            if (!go[k]) next.push_back(alive[k]);
        alive.swap(next);
    }
    return alive;
}
} // namespace proto

extern "C" {
// For every Shape of the scene with at least three candidates: both chains by the serial walk and by pop rounds (mode as above).
// stats[0] Shapes compared, [1] Shapes whose hull differs, [2] the first such Shape (or -1), [3] rounds in total, [4] most rounds of one chain,
// [5] candidates in total. Returns the number of differing Shapes.
long proto_hull_compare(void* h, int mode, long* stats) {
    using namespace oracle;
    using namespace proto;
    Scene* sc = static_cast<Scene*>(h);
    stats[0] = stats[1] = stats[3] = stats[4] = stats[5] = 0, stats[2] = -1;
    for (size_t s = 0; s < sc->shapes.size(); ++s) {
        std::vector<Safe2> pts = sc->shapes[s].hull_candidates;
        if (pts.size() < 3 || sc->shapes[s].status != CRH_OK) continue;
        std::stable_sort(pts.begin(), pts.end(), [](const Safe2& a, const Safe2& b) { return candidate_less(a, b); });
        std::vector<uint32_t> up(pts.size()), down(pts.size());
        for (uint32_t i = 0; i < pts.size(); ++i) up[i] = i, down[i] = (uint32_t)pts.size() - 1u - i;
        bool same = true;
        for (const std::vector<uint32_t>* order : {&up, &down}) {
            uint32_t rounds = 0;
            const std::vector<uint32_t> a = chain_serial(pts, *order), b = chain_rounds(pts, *order, mode, &rounds);
            stats[3] += rounds, stats[4] = std::max<long>(stats[4], rounds);
            if (a.size() != b.size()) same = false;
            else
                for (size_t k = 0; k < a.size(); ++k) // (compared by value: duplicates are bit-identical points)
                    if (pts[a[k]].v[0] != pts[b[k]].v[0] || pts[a[k]].v[1] != pts[b[k]].v[1]) same = false;
        }
        stats[0] += 1, stats[5] += (long)pts.size();
        if (!same) {
            stats[1] += 1;
            if (stats[2] < 0) stats[2] = (long)s;
        }
    }
    return stats[1];
}
// the candidates of one Shape, in emission order (x0, y0, x1, y1, ...); returns their number
uint32_t proto_hull_candidates(void* h, uint32_t s, float* out, uint32_t cap) {
    const auto& c = static_cast<Scene*>(h)->shapes[s].hull_candidates;
    for (uint32_t i = 0; i < c.size() && i < cap; ++i) out[2 * i] = c[i].v[0], out[2 * i + 1] = c[i].v[1];
    return (uint32_t)c.size();
}
}
