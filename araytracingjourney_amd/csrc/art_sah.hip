// Traversal-tree rebuild for PREFER_FAST_TRACE (vk_model.rs:968, vk_tlas_builder.rs:138): a binned surface-area-heuristic
// binary tree over the SAME leaves (one triangle each, canonical LBVH leaf order) replaces the Karras topology in the
// traversal nodes.  Results cannot change: accept() and t_eff are defined per triangle and every node box
// here is an exact min/max union of leaf boxes, so any tree over the leaves returns the same hits -- only fewer node visits.
// Host build (threads over subtrees); the canonical LBVH arrays (art_get_lbvh) stay as they are.
#include "art_internal.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace art {
namespace {

#define HIPS(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

constexpr int kBins = 32;

struct Box3 {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    void grow(const float *l, const float *h) { for (int k = 0; k < 3; k++) { lo[k] = std::fmin(lo[k], l[k]); hi[k] = std::fmax(hi[k], h[k]); } }
    void grow(const Box3 &b) { grow(b.lo, b.hi); }
    double half_area() const {
        double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        return dx < 0 ? 0.0 : dx * dy + dy * dz + dz * dx;
    }
};

struct Builder {
    const float *llo, *lhi;     // leaf boxes [T*3]
    std::vector<uint32_t> idx;  // leaf positions, partitioned in place
    std::vector<int32_t> child; // [2*NI]
    std::vector<float> nlo, nhi; // [NI*3]

    // Subtree over idx[b,e) rooted at internal node k.  Pre-order layout: the left subtree (nl leaves -> nl-1 nodes) follows its
    // parent, the right one starts at k + nl, so every node index is known before its subtree exists: subtrees build independently.
    struct Task { uint32_t b, e, k, depth; };
    // the walks' stacks are sized for the radix tree's 95 levels: past this depth the splits are medians, which bounds the
    // tree at kSahDepth + ceil(log2 T) <= 80 levels
    static constexpr uint32_t kSahDepth = 48;

    static constexpr unsigned kMaxThreads = 16;
    static constexpr uint32_t kSweep = 32;
    unsigned wide_threads = 1; // threads for the passes over one large range (the top of the tree, before subtrees run in parallel)

    // run f(chunk_begin, chunk_end, chunk_index) over [b,e) on up to wide_threads threads
    template <class F> void chunks(uint32_t b, uint32_t e, unsigned &n_chunks, F &&f) {
        const uint32_t n = e - b;
        n_chunks = (wide_threads > 1 && n >= 65536) ? wide_threads : 1;
        if (n_chunks == 1) { f(b, e, 0u); return; }
        std::vector<std::thread> pool;
        for (unsigned c = 0; c < n_chunks; c++)
            pool.emplace_back([&, c] { f(b + (uint32_t)((uint64_t)n * c / n_chunks), b + (uint32_t)((uint64_t)n * (c + 1) / n_chunks), c); });
        for (auto &t : pool) t.join();
    }

    // choose the split of idx[b,e), partition, return the middle; also writes node k's box
    uint32_t split(uint32_t b, uint32_t e, uint32_t k, uint32_t depth) {
        Box3 nb, cb;
        {
            unsigned nc = 1;
            Box3 pn[kMaxThreads], pc[kMaxThreads];
            chunks(b, e, nc, [&](uint32_t cb0, uint32_t ce, unsigned c) {
                Box3 n1, c1;
                for (uint32_t i = cb0; i < ce; i++) {
                    const float *l = llo + 3 * (size_t)idx[i], *h = lhi + 3 * (size_t)idx[i];
                    n1.grow(l, h);
                    float ct[3] = {0.5f * l[0] + 0.5f * h[0], 0.5f * l[1] + 0.5f * h[1], 0.5f * l[2] + 0.5f * h[2]};
                    c1.grow(ct, ct);
                }
                pn[c] = n1; pc[c] = c1;
            });
            for (unsigned c = 0; c < nc; c++) { nb.grow(pn[c]); cb.grow(pc[c]); }
        }
        for (int a = 0; a < 3; a++) { nlo[3 * (size_t)k + a] = nb.lo[a]; nhi[3 * (size_t)k + a] = nb.hi[a]; }
        const uint32_t n = e - b;
        if (n == 2) return b + 1;
        if (depth >= kSahDepth) return b + n / 2;
        if (n <= kSweep) { // small ranges: the exact sweep over every split of every axis (sorted by centroid)
            uint32_t ord[kSweep], best_ord[kSweep];
            double best = INFINITY; uint32_t best_left = 0;
            for (int a = 0; a < 3; a++) {
                if (!(cb.hi[a] > cb.lo[a])) continue;
                for (uint32_t i = 0; i < n; i++) ord[i] = idx[b + i];
                std::sort(ord, ord + n, [&](uint32_t x, uint32_t y) {
                    float cx = 0.5f * llo[3 * (size_t)x + a] + 0.5f * lhi[3 * (size_t)x + a], cy = 0.5f * llo[3 * (size_t)y + a] + 0.5f * lhi[3 * (size_t)y + a];
                    return cx < cy || (cx == cy && x < y);
                });
                double right_area[kSweep];
                Box3 acc;
                for (uint32_t i = n - 1; i > 0; i--) { acc.grow(llo + 3 * (size_t)ord[i], lhi + 3 * (size_t)ord[i]); right_area[i] = acc.half_area(); }
                acc = Box3{};
                for (uint32_t i = 0; i + 1 < n; i++) {
                    acc.grow(llo + 3 * (size_t)ord[i], lhi + 3 * (size_t)ord[i]);
                    double cost = acc.half_area() * (i + 1) + right_area[i + 1] * (n - 1 - i);
                    if (cost < best) { best = cost; best_left = i + 1; std::memcpy(best_ord, ord, n * sizeof(uint32_t)); }
                }
            }
            if (best_left == 0) return b + n / 2;
            for (uint32_t i = 0; i < n; i++) idx[b + i] = best_ord[i];
            return b + best_left;
        }
        // one pass fills the bins of all three axes
        struct Bins { Box3 box[3][kBins]; uint32_t cnt[3][kBins]; };
        float sc[3]; bool live[3];
        for (int a = 0; a < 3; a++) { float ext = cb.hi[a] - cb.lo[a]; live[a] = ext > 0.0f; sc[a] = live[a] ? (float)kBins / ext : 0.0f; }
        unsigned nc = 1;
        Bins first{};                       // the common case (small ranges, one thread) stays on the stack
        std::vector<Bins> more(wide_threads > 1 && n >= 65536 ? wide_threads - 1 : 0);
        auto part_of = [&](unsigned c) -> Bins & { return c == 0 ? first : more[c - 1]; };
        chunks(b, e, nc, [&](uint32_t cb0, uint32_t ce, unsigned c) {
            Bins &B = part_of(c);
            for (uint32_t i = cb0; i < ce; i++) {
                const float *l = llo + 3 * (size_t)idx[i], *h = lhi + 3 * (size_t)idx[i];
                for (int a = 0; a < 3; a++) {
                    if (!live[a]) continue;
                    int bi = (int)(((0.5f * l[a] + 0.5f * h[a]) - cb.lo[a]) * sc[a]);
                    bi = bi < 0 ? 0 : (bi >= kBins ? kBins - 1 : bi);
                    B.box[a][bi].grow(l, h); B.cnt[a][bi]++;
                }
            }
        });
        Bins &T0 = first;
        for (unsigned c = 1; c < nc; c++)
            for (int a = 0; a < 3; a++) for (int i = 0; i < kBins; i++) { T0.box[a][i].grow(more[c - 1].box[a][i]); T0.cnt[a][i] += more[c - 1].cnt[a][i]; }
        int best_axis = -1, best_bin = 0;
        double best_cost = INFINITY;
        for (int a = 0; a < 3; a++) {
            if (!live[a]) continue;
            const Box3 *bins = T0.box[a]; const uint32_t *cnt = T0.cnt[a];
            double right_area[kBins]; uint32_t right_cnt[kBins];
            Box3 acc; uint32_t c = 0;
            for (int i = kBins - 1; i > 0; i--) { acc.grow(bins[i]); c += cnt[i]; right_area[i] = acc.half_area(); right_cnt[i] = c; }
            acc = Box3{}; c = 0;
            for (int i = 0; i < kBins - 1; i++) {
                acc.grow(bins[i]); c += cnt[i];
                if (c == 0 || right_cnt[i + 1] == 0) continue;
                double cost = acc.half_area() * c + right_area[i + 1] * right_cnt[i + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = i; }
            }
        }
        if (best_axis < 0) return b + n / 2; // coincident centroids: any split is as good
        const int a = best_axis;
        const float s1 = sc[a], c0 = cb.lo[a];
        auto mid = std::partition(idx.begin() + b, idx.begin() + e, [&](uint32_t p) {
            const float *l = llo + 3 * (size_t)p, *h = lhi + 3 * (size_t)p;
            int bi = (int)(((0.5f * l[a] + 0.5f * h[a]) - c0) * s1);
            bi = bi < 0 ? 0 : (bi >= kBins ? kBins - 1 : bi);
            return bi <= best_bin;
        });
        uint32_t m = (uint32_t)(mid - idx.begin());
        if (m == b || m == e) m = b + n / 2;
        return m;
    }
    // one node; pushes the child subtrees that still need nodes
    template <class Push> void node(const Task &t, Push &&push) {
        uint32_t m = split(t.b, t.e, t.k, t.depth);
        uint32_t nl = m - t.b, nr = t.e - m;
        if (nl == 1) child[2 * (size_t)t.k] = ~(int32_t)idx[t.b];
        else { child[2 * (size_t)t.k] = (int32_t)(t.k + 1); push(Task{t.b, m, t.k + 1, t.depth + 1}); }
        if (nr == 1) child[2 * (size_t)t.k + 1] = ~(int32_t)idx[m];
        else { child[2 * (size_t)t.k + 1] = (int32_t)(t.k + nl); push(Task{m, t.e, t.k + nl, t.depth + 1}); }
    }
    void subtree(Task root) {
        std::vector<Task> st; st.push_back(root);
        while (!st.empty()) { Task t = st.back(); st.pop_back(); node(t, [&](Task c) { st.push_back(c); }); }
    }
    void build(uint32_t T, unsigned threads) {
        idx.resize(T); for (uint32_t i = 0; i < T; i++) idx[i] = i;
        child.assign((size_t)(T - 1) * 2, 0); nlo.assign((size_t)(T - 1) * 3, 0.f); nhi.assign((size_t)(T - 1) * 3, 0.f);
        // the top of the tree on this thread until there are enough independent subtrees, largest first
        std::vector<Task> open; open.push_back(Task{0, T, 0, 0});
        const size_t want = threads > 1 ? (size_t)threads * 8 : 1;
        wide_threads = threads < kMaxThreads ? threads : kMaxThreads;
        // Rounds: every open range of >= 4096 leaves is split once per round.  While there are few of them each split runs its passes on
        // all threads (chunks()); from four ranges on the ranges themselves are the parallel work, one thread each (the serial middle of the
        // tree -- ranges between 4 096 and 65 535 leaves split one after another -- used to be about half of the build).
        while (threads > 1 && open.size() < want) {
            std::vector<Task> big, rest;
            for (const Task &t : open) (t.e - t.b >= 4096 ? big : rest).push_back(t);
            if (big.empty()) break;
            std::vector<std::vector<Task>> out(big.size());
            if (big.size() < 4) {
                wide_threads = threads < kMaxThreads ? threads : kMaxThreads;
                for (size_t i = 0; i < big.size(); i++) node(big[i], [&](Task c) { out[i].push_back(c); });
            } else {
                wide_threads = 1;
                std::atomic<size_t> next{0};
                std::vector<std::thread> pool;
                for (unsigned w = 0; w < threads && w < big.size(); w++)
                    pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < big.size();) node(big[i], [&, i](Task c) { out[i].push_back(c); }); });
                for (auto &th : pool) th.join();
            }
            open = rest;
            for (const auto &v : out) open.insert(open.end(), v.begin(), v.end());
        }
        wide_threads = 1;
        if (threads <= 1 || open.size() < 2) { for (const Task &t : open) subtree(t); return; }
        std::sort(open.begin(), open.end(), [](const Task &x, const Task &y) { return x.e - x.b > y.e - y.b; });
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < threads; w++)
            pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < open.size();) subtree(open[i]); });
        for (auto &th : pool) th.join();
    }
};

} // namespace

hipError_t sah_build(Lbvh &l, uint32_t T, hipStream_t s) {
    if (T < 3) return hipSuccess; // one node at most: nothing to choose
    const uint32_t NI = T - 1;
    const bool log = (l.log & 1u) != 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto t0 = now();
    std::vector<float> llo((size_t)T * 3), lhi((size_t)T * 3);
    HIPS(hipStreamSynchronize(s));
    HIPS(hipMemcpy(llo.data(), l.leaf_lo, (size_t)T * 12, hipMemcpyDeviceToHost));
    HIPS(hipMemcpy(lhi.data(), l.leaf_hi, (size_t)T * 12, hipMemcpyDeviceToHost));
    auto t1 = now();
    Builder B; B.llo = llo.data(); B.lhi = lhi.data();
    unsigned hw = std::thread::hardware_concurrency();
    B.build(T, T < 20000 ? 1u : std::min(hw ? hw : 1u, 16u));
    auto t2 = now();
    std::vector<DevNode> nodes(NI);
    auto box = [&](int32_t ref, const float *&lo, const float *&hi) {
        if (ref < 0) { lo = llo.data() + 3 * (size_t)(~ref); hi = lhi.data() + 3 * (size_t)(~ref); }
        else { lo = B.nlo.data() + 3 * (size_t)ref; hi = B.nhi.data() + 3 * (size_t)ref; }
    };
    auto emit = [&](uint32_t n0, uint32_t n1) { for (uint32_t n = n0; n < n1; n++) { // the layout k_emit_nodes writes
        int32_t c0 = B.child[2 * (size_t)n], c1 = B.child[2 * (size_t)n + 1];
        const float *l0, *h0, *l1, *h1; box(c0, l0, h0); box(c1, l1, h1);
        DevNode &d = nodes[n];
        d.q[0] = make_float4(l0[0], l0[1], l0[2], h0[0]);
        d.q[1] = make_float4(h0[1], h0[2], l1[0], l1[1]);
        d.q[2] = make_float4(l1[2], h1[0], h1[1], h1[2]);
        int32_t cc[2] = {c0, c1}; float cf[2]; std::memcpy(cf, cc, 8);
        d.q[3] = make_float4(cf[0], cf[1], 0.f, 0.f);
    } };
    {
        unsigned nt = NI >= 65536 ? std::min(hw ? hw : 1u, 16u) : 1u;
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < nt; w++) pool.emplace_back(emit, (uint32_t)((uint64_t)NI * w / nt), (uint32_t)((uint64_t)NI * (w + 1) / nt));
        for (auto &th : pool) th.join();
    }
    auto t3 = now();
    HIPS(hipMemcpy(l.nodes, nodes.data(), (size_t)NI * sizeof(DevNode), hipMemcpyHostToDevice));
    HIPS(lbvh_claim_trav(l, NI));
    HIPS(hipMemcpy(l.trav_child, B.child.data(), (size_t)NI * 8, hipMemcpyHostToDevice));
    HIPS(hipMemcpy(l.trav_lo, B.nlo.data(), (size_t)NI * 12, hipMemcpyHostToDevice));
    HIPS(hipMemcpy(l.trav_hi, B.nhi.data(), (size_t)NI * 12, hipMemcpyHostToDevice));
    if (log) std::fprintf(stderr, "[art] sah_build %u leaves: wait + read-back %.1f ms, build %.1f, node records %.1f, upload %.1f\n", T, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, now()));
    return hipSuccess;
}

} // namespace art
