// The PREFER_FAST_TRACE traversal tree built on the device: a binned surface-area heuristic over the LBVH's leaves (one triangle per leaf, pre-order node
// numbering: the left subtree follows its parent, the right one starts nl nodes on), level by level over ALL open ranges at once, each range by the kernel
// that fits its size.  Every node box is an exact min/max union of leaf boxes (unions go through an order-preserving float <-> uint key, so the atomics are
// integer min / max), hence frames cannot depend on which builder ran.
//
//   a range of more than kMid leaves:   k_bin     windows of positions -> LDS histograms (64 bins x 3 axes over the range's domain) -> one atomic per touched word
//                                       k_choose  a WAVE per range: a lane per bin, two scans, every plane's cost at once, wave arg-min
//                                       k_flags + scan + k_scatter + k_leaf_refs: stable partition of every open range at once (these levels only)
//   kSmall < leaves <= kMid:            k_mid     a block per range: bins in LDS, the plane chosen there; once no larger range is open, the level's only kernel --
//                                                 it then partitions its range in place and writes the one-leaf sides
//   2 .. kSmall leaves:                 k_small   (once, after the levels) a thread per range: the whole subtree by the exact sweep
// A child's binning domain comes from the parent's bins (no centroid pass below the root); the children's places in the next level's / the small ranges' lists
// are handed out a batch of ranges at a time; the levels are launched four at a time between fences; all temporaries come from the context's arena.
// Round 3b (profiles/README.md): 11.8 -> 3.3 ms for config 2's 262 816 triangles, 62.5 -> 11.1 ms for config 4's 2.8 M, at +0.8 % of the ray rate.  The levels had
// been one leaf-by-leaf pass: 21 atomics a leaf for the bins and 6 for the centroid bounds, 48 of config 4's 62 ms; bins on ONE axis (the longest side: pbrt's
// rule) cost config 4 8.5 % of its ray rate and were dropped.
#include "art_internal.h"
#include <rocprim/rocprim.hpp>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace art {
namespace {

#define HIPQ(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

#ifndef ART_SAH_BINS
#define ART_SAH_BINS 64
#endif
constexpr int kBins = ART_SAH_BINS;
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kSahDepth = 48; // past it ranges are halved: the tree stays within kSahDepth + log2(T) levels (the walks' stacks)
constexpr uint32_t kBlockB = 256;
#ifndef ART_SAH_SMALL
#define ART_SAH_SMALL 16
#endif
constexpr uint32_t kSmall = ART_SAH_SMALL;    // a range of at most this many leaves is finished by one thread (k_small)
#ifndef ART_SAH_MID
#define ART_SAH_MID 4096
#endif
constexpr uint32_t kMid = ART_SAH_MID;    // a range of at most this many leaves (and more than kSmall) is binned and split by one block in its LDS (k_mid); larger ones leaf by leaf (k_bin + k_choose)
constexpr uint32_t kSmallDepth = 10; // ... whose exact sweep may chain at most this deep before it halves (a degenerate fan of 16 leaves would be 15 levels)

#ifndef ART_SAH_AXES
#define ART_SAH_AXES 3   // axes a range is binned on: 3 = all (21 atomics a leaf), 1 = the longest side of its box only (7: the rule of pbrt's builder)
#endif
constexpr int kAxes = ART_SAH_AXES;
struct Range { uint32_t b, e, k, depth; float lo[3], hi[3]; uint32_t axis, pad; }; // leaves idx[b, e) -> internal node k; binned over its own box [lo, hi] (kAxes == 1: on `axis`, the longest side)
struct Split { uint32_t axis, bin, nl, left, right; };   // bin kNone: by position (median); left / right: the child ranges' ids in the next level, or kNone (a leaf, or a small range)
__device__ __forceinline__ uint32_t bin_slot(const Range &R) { return R.b / kMid; }

__device__ __forceinline__ uint32_t fkey(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float fkey_inv(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }
__device__ __forceinline__ float centroid(const float *lo, const float *hi, uint32_t leaf, int a) { return 0.5f * lo[3 * (size_t)leaf + a] + 0.5f * hi[3 * (size_t)leaf + a]; }
__device__ __forceinline__ int bin_of(float c, float c0, float sc) { return (int)fminf(fmaxf((c - c0) * sc, 0.0f), (float)(kBins - 1)); }   // (clamped as a float: the conversion of a NaN or of 1e30 is not defined)
__device__ __forceinline__ int bin_in(const Range &R, int a, const float *lo, const float *hi, uint32_t leaf) { return R.hi[a] > R.lo[a] ? bin_of(centroid(lo, hi, leaf, a), R.lo[a], (float)kBins / (R.hi[a] - R.lo[a])) : 0; }   // a flat side: everything in bin 0, never chosen

// bins: [slot][axis][bin][7] = lo xyz keys (initialised to ~0), hi xyz keys (0), count.  Only ranges of more than kMid leaves have bins in memory, and such ranges are
// disjoint intervals longer than kMid: b / kMid is a slot of its own for each of them (T / kMid + 1 slots: 3.7 MB for config 4 instead of 2688 bytes for every open range)
// (the number of open ranges of a level is read from the device -- n_dev, the counter the level before filled -- so that a level can be launched before the
// host knows it: grid-stride loops over whatever the launch was given)
__global__ __launch_bounds__(kBlockB) void k_init_level(const uint32_t *__restrict__ n_dev, const Range *__restrict__ ranges, uint32_t *bins) {   // a block per range; only the ranges k_bin fills
    const uint32_t n = *n_dev;
    for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
        if (ranges[r].e - ranges[r].b <= kMid) continue;
        for (uint32_t w = threadIdx.x; w < kAxes * kBins * 7; w += kBlockB) bins[(size_t)bin_slot(ranges[r]) * kAxes * kBins * 7 + w] = (w % 7) < 3 ? 0xFFFFFFFFu : 0u;
    }
}
struct Box { float lo[3], hi[3]; };
__device__ __forceinline__ void box_empty(Box &b) { for (int k = 0; k < 3; k++) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; } }
__device__ __forceinline__ void box_grow(Box &b, const uint32_t *w) { for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], fkey_inv(w[k])); b.hi[k] = fmaxf(b.hi[k], fkey_inv(w[3 + k])); } }
__device__ __forceinline__ void box_grow(Box &b, const float *lo, const float *hi) { for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], lo[k]); b.hi[k] = fmaxf(b.hi[k], hi[k]); } }
__device__ __forceinline__ double half_area(const Box &b) { double dx = (double)b.hi[0] - b.lo[0], dy = (double)b.hi[1] - b.lo[1], dz = (double)b.hi[2] - b.lo[2]; return dx < 0 ? 0.0 : dx * dy + dy * dz + dz * dx; }
// a range's domain (and, when only one axis is binned, which: the longest side); on a flat side everything lands in bin 0 and the side is never chosen
__device__ __forceinline__ void set_domain(Range &R, const Box &b) {
    const float ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
    const int a = ex >= ey ? (ex >= ez ? 0 : 2) : (ey >= ez ? 1 : 2);
    R.axis = (uint32_t)a; R.pad = 0;
    for (int k = 0; k < 3; k++) { R.lo[k] = b.lo[k]; R.hi[k] = b.hi[k]; }
}
__global__ void k_root_range(uint32_t T, const uint32_t *__restrict__ box, Range *ranges) {
    Box b; for (int a = 0; a < 3; a++) { b.lo[a] = fkey_inv(box[a]); b.hi[a] = fkey_inv(box[3 + a]); }
    Range R{0, T, 0, 0, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, 0, 0};
    set_domain(R, b);
    ranges[0] = R;
}
// The bins of the ranges of more than kMid leaves, leaf by leaf.  A block takes a window of kWin positions; a window is shorter than such a range, so it holds leaves
// of at most two of them (the one that ends in it, the one that begins): each gets the block's LDS histogram in turn, and what leaves the CU is one atomic per
// touched word and window -- leaf by leaf straight to memory, 262 k leaves on the root's words took 3.3 ms of serialised atomics.
constexpr uint32_t kWin = 2048, kWinItems = kWin / kBlockB;
static_assert(kWin <= kMid, "k_bin: a window holds at most two large ranges");
__global__ __launch_bounds__(kBlockB) void k_bin(uint32_t T, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ range_of, const float *__restrict__ lo, const float *__restrict__ hi,
                                                 const Range *__restrict__ ranges, uint32_t *bins) {
    __shared__ uint32_t s_lo, s_hi, s_bins[kAxes * kBins * 7];
    const uint32_t w0 = blockIdx.x * kWin;
    if (threadIdx.x == 0) { s_lo = 0xFFFFFFFFu; s_hi = 0u; }
    __syncthreads();
    uint32_t rr[kWinItems], mn = 0xFFFFFFFFu, mx = 0u;
    for (uint32_t it = 0; it < kWinItems; it++) {
        const uint32_t i = w0 + it * kBlockB + threadIdx.x;
        uint32_t r = i < T ? range_of[i] : kNone;
        if (r != kNone && ranges[r].e - ranges[r].b <= kMid) r = kNone;   // k_mid's
        rr[it] = r;
        if (r != kNone) { mn = min(mn, r); mx = max(mx, r); }
    }
    if (mn != 0xFFFFFFFFu) { atomicMin(&s_lo, mn); atomicMax(&s_hi, mx); }
    __syncthreads();
    if (s_lo == 0xFFFFFFFFu) return;                      // no large range in this window
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t target = pass == 0 ? s_lo : s_hi;
        if (pass == 1 && s_hi == s_lo) break;
        for (uint32_t w = threadIdx.x; w < kAxes * kBins * 7; w += kBlockB) s_bins[w] = (w % 7) < 3 ? 0xFFFFFFFFu : 0u;
        __syncthreads();
        const Range R = ranges[target];
        for (uint32_t it = 0; it < kWinItems; it++) {
            if (rr[it] != target) continue;
            const uint32_t leaf = idx[w0 + it * kBlockB + threadIdx.x];
            uint32_t kl[3], kh[3];
            for (int k = 0; k < 3; k++) { kl[k] = fkey(lo[3 * (size_t)leaf + k]); kh[k] = fkey(hi[3 * (size_t)leaf + k]); }
            for (int j = 0; j < kAxes; j++) {
                const int a = kAxes == 1 ? (int)R.axis : j;
                uint32_t *w = s_bins + ((size_t)j * kBins + bin_in(R, a, lo, hi, leaf)) * 7;
                for (int k = 0; k < 3; k++) { atomicMin(&w[k], kl[k]); atomicMax(&w[3 + k], kh[k]); }
                atomicAdd(&w[6], 1u);
            }
        }
        __syncthreads();
        uint32_t *g = bins + (size_t)bin_slot(R) * kAxes * kBins * 7;
        for (uint32_t w = threadIdx.x; w < kAxes * kBins * 7; w += kBlockB) {
            const uint32_t k = w % 7, v = s_bins[w];
            if (s_bins[w - k + 6] == 0) continue;             // empty bin
            if (k < 3) atomicMin(&g[w], v); else if (k < 6) atomicMax(&g[w], v); else atomicAdd(&g[w], v);
        }
        __syncthreads();
    }
}
struct SmallRange { uint32_t b, e, k, depth; };
__device__ __forceinline__ Box box_shfl(const Box &b, int src) { Box o; for (int k = 0; k < 3; k++) { o.lo[k] = __shfl(b.lo[k], src); o.hi[k] = __shfl(b.hi[k], src); } return o; }
// One WAVE per range, a lane per bin (kBins <= 64): the boxes and counts of the bins in front of / behind every plane are two scans across the wave, the costs of all the
// planes of an axis come out at once, the best of them by a wave-wide arg-min -- the thread that used to walk the 3 x 32 bins of a range alone took 55 us whatever the level
// (the root: ONE thread busy on the whole chip), 213 us at 64 bins: 4.3 of config 2's 12.7 ms.  Same costs in the same doubles, ties to the lower (axis, bin): the same tree.
static_assert(kBins <= 64, "k_choose: a lane per bin");
struct LevelOut { Split *splits; Range *next; uint32_t *n_next; SmallRange *small; uint32_t *n_small; int32_t *child; float *nlo, *nhi; uint32_t *n_big; };   // n_big: how many of the next level's ranges have more than kMid leaves
// What a range's split leaves to be written once its children have their places in the next level's / the small ranges' lists.  The places are handed out by ONE atomic per
// list for a whole batch of ranges (commit_batch): an atomic per child on the two counters was the deep levels' time -- 43 k ranges x 2 returning atomics on one word, at
// the ~90 per microsecond a word takes (MI355X_MICROARCH.md), is the 1.3 ms a level that k_mid measured.
struct Pending { Split S; Range L, Rr; uint32_t r; uint32_t kind_l, kind_r; };   // kind: 0 a leaf (or nothing), 1 a small range, 2 a range of the next level; r kNone: empty
// called by all 64 lanes of a wave; wr: the range's bins (global memory, or the LDS of k_mid); lane 0 fills *out
__device__ __forceinline__ void choose_range(const Range &R, uint32_t r, const uint32_t *wr, int lane, const LevelOut &o, Pending *out) {
    int32_t *child = o.child; float *nlo = o.nlo, *nhi = o.nhi;
    const uint32_t n = R.e - R.b;
    const bool sah = n > 2 && R.depth < kSahDepth;
    double best_cost = INFINITY; int best_key = 0x7FFFFFFF; uint32_t best_left = 0;
    Box node, lbs, rbs;   // the node's box; the boxes on either side of this lane's best plane so far
    box_empty(node); box_empty(lbs); box_empty(rbs);
    for (int j = 0; j < kAxes; j++) {
        const int a = kAxes == 1 ? (int)R.axis : j;
        Box p; box_empty(p); uint32_t cp = 0;
        if (lane < kBins) { const uint32_t *w = wr + ((size_t)j * kBins + lane) * 7; cp = w[6]; if (cp) box_grow(p, w); }
        Box q = p; uint32_t cq = cp;
        for (int off = 1; off < 64; off <<= 1) {   // p, cp: bins 0 .. lane;  q, cq: bins lane .. last
            Box u, d; uint32_t cu = (uint32_t)__shfl_up((int)cp, off), cd = (uint32_t)__shfl_down((int)cq, off);
            for (int k = 0; k < 3; k++) { u.lo[k] = __shfl_up(p.lo[k], off); u.hi[k] = __shfl_up(p.hi[k], off); d.lo[k] = __shfl_down(q.lo[k], off); d.hi[k] = __shfl_down(q.hi[k], off); }
            if (lane >= off) { for (int k = 0; k < 3; k++) { p.lo[k] = fminf(p.lo[k], u.lo[k]); p.hi[k] = fmaxf(p.hi[k], u.hi[k]); } cp += cu; }
            if (lane + off < 64) { for (int k = 0; k < 3; k++) { q.lo[k] = fminf(q.lo[k], d.lo[k]); q.hi[k] = fmaxf(q.hi[k], d.hi[k]); } cq += cd; }
        }
        if (j == 0) node = box_shfl(p, 63);   // every leaf is in some bin of the first binned axis
        // the plane behind bin `lane`: left = bins 0 .. lane, right = bins lane + 1 .. last
        Box qr; uint32_t cr = (uint32_t)__shfl_down((int)cq, 1);
        for (int k = 0; k < 3; k++) { qr.lo[k] = __shfl_down(q.lo[k], 1); qr.hi[k] = __shfl_down(q.hi[k], 1); }
        const bool ok = sah && R.hi[a] > R.lo[a] && lane < kBins - 1 && cp != 0 && cr != 0;
        const double cost = ok ? half_area(p) * cp + half_area(qr) * cr : (double)INFINITY;
        if (cost < best_cost) { best_cost = cost; best_key = j * kBins + lane; best_left = cp; lbs = p; rbs = qr; }
    }
    // the wave's best plane: lowest cost, ties to the lower (axis, bin)
    double wc = best_cost; int wk = best_key;
    for (int off = 32; off >= 1; off >>= 1) {
        const double oc = __shfl_xor(wc, off); const int ok_ = __shfl_xor(wk, off);
        if (oc < wc || (oc == wc && ok_ < wk)) { wc = oc; wk = ok_; }
    }
    const bool split = wc < (double)INFINITY;
    const int best_j = split ? wk / kBins : 0, best_bin = split ? wk % kBins : -1;
    Box lb = box_shfl(lbs, split ? best_bin : 0), rb = box_shfl(rbs, split ? best_bin : 0);
    const uint32_t left_n = (uint32_t)__shfl((int)best_left, split ? best_bin : 0);
    if (lane != 0) return;
    for (int k = 0; k < 3; k++) { nlo[3 * (size_t)R.k + k] = node.lo[k]; nhi[3 * (size_t)R.k + k] = node.hi[k]; }
    Split S;
    if (!split) { S.axis = 0; S.bin = kNone; S.nl = n / 2; }       // two leaves, coincident centroids, or past the depth guard
    else { S.axis = kAxes == 1 ? R.axis : (uint32_t)best_j; S.bin = (uint32_t)best_bin; S.nl = left_n; }
    const uint32_t nl = S.nl, nr = n - nl;
    S.left = kNone; S.right = kNone;
    if (split) {
        // A child's domain is where its leaves' CENTROIDS can be, not where their boxes reach (a wall's two triangles span the node; their centroids do not): inside
        // the parent's domain, on the split axis on the child's side of the plane (a centroid in bin i lies within a rounding of the bin's edges, hence the margin of
        // half a bin), and inside the child's own box.  The centroid bounds a separate pass over the leaves used to make (6 atomics a leaf and level) are not needed:
        // with 64 bins the looser domain costs nothing (config 4: 17 270 Mray/s against 17 140 with 32 bins over exact centroid bounds; 16 750 with 32 bins over these).
        const int a = (int)S.axis;
        const float w = (R.hi[a] - R.lo[a]) * (1.0f / kBins), plane = R.lo[a] + w * (float)(best_bin + 1);
        for (int k = 0; k < 3; k++) {
            const float dl = R.lo[k], dh = R.hi[k];
            lb.lo[k] = fmaxf(lb.lo[k], dl); lb.hi[k] = fminf(lb.hi[k], k == a ? fminf(dh, plane + 0.5f * w) : dh);
            rb.lo[k] = fmaxf(rb.lo[k], k == a ? fmaxf(dl, plane - 0.5f * w) : dl); rb.hi[k] = fminf(rb.hi[k], dh);
        }
    } else for (int k = 0; k < 3; k++) { lb.lo[k] = rb.lo[k] = fmaxf(node.lo[k], R.lo[k]); lb.hi[k] = rb.hi[k] = fminf(node.hi[k], R.hi[k]); }   // a range split by position bins its halves over its own box
    Pending P; P.r = r; P.kind_l = 0; P.kind_r = 0;
    P.L = Range{R.b, R.b + nl, R.k + 1, R.depth + 1, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, 0, 0};
    P.Rr = Range{R.b + nl, R.e, R.k + nl, R.depth + 1, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, 0, 0};
    if (nl > 1) { child[2 * (size_t)R.k] = (int32_t)(R.k + 1); if (nl <= kSmall) P.kind_l = 1; else { P.kind_l = 2; set_domain(P.L, lb); } }
    if (nr > 1) { child[2 * (size_t)R.k + 1] = (int32_t)(R.k + nl); if (nr <= kSmall) P.kind_r = 1; else { P.kind_r = 2; set_domain(P.Rr, rb); } }
    P.S = S;
    *out = P;
}
// one thread: the places of a batch's children (one atomic per list), then the records
__device__ __forceinline__ void commit_batch(Pending *p, uint32_t count, const LevelOut &o) {
    uint32_t cn = 0, cs = 0;
    for (uint32_t j = 0; j < count; j++) if (p[j].r != kNone) { cn += (p[j].kind_l == 2) + (p[j].kind_r == 2); cs += (p[j].kind_l == 1) + (p[j].kind_r == 1); }
    uint32_t bn = cn ? atomicAdd(o.n_next, cn) : 0u, bs = cs ? atomicAdd(o.n_small, cs) : 0u, big = 0;
    for (uint32_t j = 0; j < count; j++) if (p[j].r != kNone) { big += (p[j].kind_l == 2 && p[j].L.e - p[j].L.b > kMid) + (p[j].kind_r == 2 && p[j].Rr.e - p[j].Rr.b > kMid); }
    if (big) atomicAdd(o.n_big, big);
    for (uint32_t j = 0; j < count; j++) {
        if (p[j].r == kNone) continue;
        Split S = p[j].S;
        if (p[j].kind_l == 2) { S.left = bn; o.next[bn++] = p[j].L; } else if (p[j].kind_l == 1) o.small[bs++] = SmallRange{p[j].L.b, p[j].L.e, p[j].L.k, p[j].L.depth};
        if (p[j].kind_r == 2) { S.right = bn; o.next[bn++] = p[j].Rr; } else if (p[j].kind_r == 1) o.small[bs++] = SmallRange{p[j].Rr.b, p[j].Rr.e, p[j].Rr.k, p[j].Rr.depth};
        if (o.splits) o.splits[p[j].r] = S;
    }
}
// ranges of more than kMid leaves: their bins were made leaf by leaf (k_bin)
__global__ __launch_bounds__(64) void k_choose(const uint32_t *__restrict__ n_dev, const Range *__restrict__ ranges, const uint32_t *__restrict__ bins, LevelOut o) {
    __shared__ Pending s_p;
    const uint32_t n_ranges = *n_dev;
    for (uint32_t r = blockIdx.x; r < n_ranges; r += gridDim.x) {
        const Range R = ranges[r];
        if (R.e - R.b <= kMid) continue;
        choose_range(R, r, bins + (size_t)bin_slot(R) * kAxes * kBins * 7, (int)threadIdx.x, o, &s_p);
        if (threadIdx.x == 0) commit_batch(&s_p, 1, o);   // (few of these: at most T / kMid a level)
    }
}
// Ranges of kSmall < n <= kMid leaves, a BLOCK per range: its leaves are an interval of positions, so the block walks them, bins them in its LDS and chooses the plane
// there -- no bins in memory, no atomics outside the CU.  (Leaf by leaf these levels were the build: ten levels of config 4 at 1 - 2.4 ms of k_bin each, 59 M atomics a level
// scattered over 230 MB of bins, and another millisecond of k_choose reading them back.)
constexpr uint32_t kBatch = 8;   // ranges a block of k_mid takes before it asks the two counters for their children's places
// FULL: the level's only kernel once no range of more than kMid leaves is open (the host learns that at a fence) -- the block also PARTITIONS its range, in place (the
// leaves staged in LDS: a stable split by the chosen plane) and writes the references of one-leaf sides: no flags, no scan over T, no scatter, no range-of-position array,
// two launches a level instead of a dozen.  Not FULL: the level's other kernels do that for every open range at once (k_flags, the scan, k_scatter, k_leaf_refs).
template <bool FULL> __global__ __launch_bounds__(kBlockB) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_mid(const uint32_t *__restrict__ n_dev, const Range *__restrict__ ranges, uint32_t *idx, const float *__restrict__ lo, const float *__restrict__ hi, LevelOut o) {
    __shared__ uint32_t s_bins[kAxes * kBins * 7];
    __shared__ Pending s_p[kBatch];
    __shared__ uint32_t s_out[FULL ? kMid : 1], s_wave[kBlockB / 64], s_base;
    const uint32_t n_ranges = *n_dev;
    // a batch is walked range after range: its length is the level's latency when the ranges are few (8 x ~12 us against a chip two thirds idle), the counters' relief when
    // they are many (one returning atomic per list and batch; a word takes ~90 a microsecond)
    const uint32_t batch = n_ranges <= 512 ? 1u : (n_ranges <= 4096 ? 2u : (n_ranges <= 16384 ? 4u : kBatch));
    for (uint32_t r0 = blockIdx.x * batch; r0 < n_ranges; r0 += gridDim.x * batch) {
        for (uint32_t j = 0; j < batch; j++) {
            const uint32_t r = r0 + j;
            if (threadIdx.x == 0) s_p[j].r = kNone;
            if (r >= n_ranges) continue;            // (block-uniform)
            const Range R = ranges[r];
            if (R.e - R.b > kMid) continue;         // (FULL: there is none)
            for (uint32_t w = threadIdx.x; w < kAxes * kBins * 7; w += kBlockB) s_bins[w] = (w % 7) < 3 ? 0xFFFFFFFFu : 0u;
            __syncthreads();
            for (uint32_t i = R.b + threadIdx.x; i < R.e; i += kBlockB) {
                const uint32_t leaf = idx[i];
                uint32_t kl[3], kh[3];
                for (int k = 0; k < 3; k++) { kl[k] = fkey(lo[3 * (size_t)leaf + k]); kh[k] = fkey(hi[3 * (size_t)leaf + k]); }
                for (int jj = 0; jj < kAxes; jj++) {
                    const int a = kAxes == 1 ? (int)R.axis : jj;
                    uint32_t *w = s_bins + ((size_t)jj * kBins + bin_in(R, a, lo, hi, leaf)) * 7;
                    for (int k = 0; k < 3; k++) { atomicMin(&w[k], kl[k]); atomicMax(&w[3 + k], kh[k]); }
                    atomicAdd(&w[6], 1u);
                }
            }
            __syncthreads();
            if (threadIdx.x < 64) choose_range(R, r, s_bins, (int)threadIdx.x, o, &s_p[j]);
            if (FULL && threadIdx.x == 0) s_base = 0;
            __syncthreads();
            if (FULL) {
                const Split S = s_p[j].S;
                const uint32_t n = R.e - R.b, nl = S.nl;
                for (uint32_t c0 = 0; c0 < n; c0 += kBlockB) {
                    const uint32_t i = c0 + threadIdx.x;
                    const bool valid = i < n;
                    const uint32_t leaf = valid ? idx[R.b + i] : 0u;
                    const bool f = valid && (S.bin == kNone ? i < nl : bin_in(R, (int)S.axis, lo, hi, leaf) <= (int)S.bin);
                    const uint64_t m = __ballot(f);
                    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
                    if (lane == 0) s_wave[wv] = (uint32_t)__popcll(m);
                    __syncthreads();
                    uint32_t before = s_base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));   // left-going leaves in front of i
                    for (uint32_t w = 0; w < wv; w++) before += s_wave[w];
                    if (valid) s_out[f ? before : nl + (i - before)] = leaf;
                    __syncthreads();
                    if (threadIdx.x == 0) { uint32_t t = 0; for (uint32_t w = 0; w < kBlockB / 64; w++) t += s_wave[w]; s_base += t; }
                    __syncthreads();
                }
                for (uint32_t i = threadIdx.x; i < n; i += kBlockB) idx[R.b + i] = s_out[i];
                if (threadIdx.x == 0) {
                    if (nl == 1) o.child[2 * (size_t)R.k] = ~(int32_t)s_out[0];
                    if (n - nl == 1) o.child[2 * (size_t)R.k + 1] = ~(int32_t)s_out[nl];
                }
                __syncthreads();
            }
        }
        if (threadIdx.x == 0) commit_batch(s_p, batch, o);
        __syncthreads();
    }
}
__global__ void k_level_reset(uint32_t *n_cnt, int next) { n_cnt[next] = 0; n_cnt[4 + next] = 0; }
__global__ __launch_bounds__(kBlockB) void k_flags(uint32_t T, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ range_of, const float *__restrict__ lo, const float *__restrict__ hi,
                                                   const Range *__restrict__ ranges, const Split *__restrict__ splits, uint32_t *flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    uint32_t r = range_of[i], f = 0;
    if (r != kNone) {
        const Split S = splits[r];
        const Range R = ranges[r];
        if (S.bin == kNone) f = (i - R.b) < S.nl;
        else f = bin_in(R, (int)S.axis, lo, hi, idx[i]) <= (int)S.bin;
    }
    flags[i] = f;
}
__global__ __launch_bounds__(kBlockB) void k_scatter(uint32_t T, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ range_of, const Range *__restrict__ ranges, const Split *__restrict__ splits,
                                                     const uint32_t *__restrict__ flags, const uint32_t *__restrict__ scan, uint32_t *idx2, uint32_t *range_of2) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    uint32_t r = range_of[i];
    if (r == kNone) { idx2[i] = idx[i]; range_of2[i] = kNone; return; }
    const Range R = ranges[r]; const Split S = splits[r];
    uint32_t left_before = scan[i] - scan[R.b];                        // left-going leaves of this range in front of i
    uint32_t dst = flags[i] ? R.b + left_before : R.b + S.nl + ((i - R.b) - left_before);
    idx2[dst] = idx[i];
    range_of2[dst] = flags[i] ? S.left : S.right;                      // kNone: a leaf, or a small range (the position keeps its leaf from here on)
}
__global__ void k_leaf_refs(const uint32_t *__restrict__ n_dev, const Range *__restrict__ ranges, const Split *__restrict__ splits, const uint32_t *__restrict__ idx2, int32_t *child) {
    const uint32_t n_ranges = *n_dev;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_ranges; r += gridDim.x * blockDim.x) {
        const Range R = ranges[r]; const Split S = splits[r];
        if (S.nl == 1) child[2 * (size_t)R.k] = ~(int32_t)idx2[R.b];
        if (R.e - R.b - S.nl == 1) child[2 * (size_t)R.k + 1] = ~(int32_t)idx2[R.b + S.nl];
    }
}
__global__ void k_iota(uint32_t T, uint32_t *idx, uint32_t *range_of, uint32_t first_range) { uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < T) { idx[i] = i; range_of[i] = first_range; } }

// One thread finishes a small range: the whole subtree under node k over the <= kSmall leaves idx[b, e), by the exact sweep -- on each axis the leaves sorted by
// centroid, every position a candidate, cost = area(left) * n_left + area(right) * n_right -- iteratively with a stack of sub-ranges (pre-order numbering as above).
// The leaves' boxes are copied once and stay where they are: what is sorted is a byte per leaf (its place in the order of the sub-range); a sub-range left sorted on the
// winning axis is its own partition.  The boxes, the order and the sweep's right-hand areas live in LDS, [index][lane] (the bank is the lane's, whatever the index: no
// conflicts): as a thread's own arrays they were scratch memory behind per-lane indices, and the kernel waited on it -- 2.4 ms for config 4's 256 k small ranges.
static_assert(kSmall <= 255, "k_small: a byte per leaf");
struct SmallLds { float box[kSmall * 6][64]; float right_area[kSmall][64]; uint8_t ord[kSmall][64]; };
#define SBOX(i, k) (L.box[(i) * 6 + (k)][lane])   /* k: 0..2 lo, 3..5 hi */
__device__ __forceinline__ float small_centroid(const SmallLds &L, uint32_t lane, uint32_t o, int a) { return 0.5f * SBOX(o, a) + 0.5f * SBOX(o, 3 + a); }
__device__ __forceinline__ void small_sort(SmallLds &L, uint32_t lane, uint32_t b, uint32_t n, int a) {   // insertion sort of ord[b, b + n) by centroid on axis a, stable
    for (uint32_t i = 1; i < n; i++) {
        const uint8_t o = L.ord[b + i][lane];
        const float ck = small_centroid(L, lane, o, a);
        uint32_t j = i;
        while (j > 0) { const uint8_t p = L.ord[b + j - 1][lane]; if (!(small_centroid(L, lane, p, a) > ck)) break; L.ord[b + j][lane] = p; j--; }
        L.ord[b + j][lane] = o;
    }
}
__device__ __forceinline__ void small_grow(Box &acc, const SmallLds &L, uint32_t lane, uint32_t o) { for (int k = 0; k < 3; k++) { acc.lo[k] = fminf(acc.lo[k], SBOX(o, k)); acc.hi[k] = fmaxf(acc.hi[k], SBOX(o, 3 + k)); } }
__global__ __launch_bounds__(64) void k_small(const uint32_t *__restrict__ n_dev, const SmallRange *__restrict__ small, const uint32_t *__restrict__ idx, const float *__restrict__ lo, const float *__restrict__ hi,
                                              int32_t *child, float *nlo, float *nhi) {
    __shared__ SmallLds L;
    const uint32_t lane = threadIdx.x;
    const uint32_t n_small = *n_dev;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_small; s += gridDim.x * blockDim.x) {
        const SmallRange SR = small[s];
        const uint32_t n0 = SR.e - SR.b;
        uint32_t leaf[kSmall];
        for (uint32_t i = 0; i < n0; i++) { const uint32_t l = idx[SR.b + i]; leaf[i] = l; L.ord[i][lane] = (uint8_t)i; for (int a = 0; a < 3; a++) { SBOX(i, a) = lo[3 * (size_t)l + a]; SBOX(i, 3 + a) = hi[3 * (size_t)l + a]; } }
        struct Sub { uint8_t b, n, depth; uint32_t k; } stack[kSmall];
        int sp = 0;
        stack[sp++] = Sub{0, (uint8_t)n0, 0, SR.k};
        while (sp > 0) {
            const Sub U = stack[--sp];
            const uint32_t b = U.b, n = U.n;
            Box node; box_empty(node);
            for (uint32_t i = 0; i < n; i++) small_grow(node, L, lane, L.ord[b + i][lane]);
            for (int a = 0; a < 3; a++) { nlo[3 * (size_t)U.k + a] = node.lo[a]; nhi[3 * (size_t)U.k + a] = node.hi[a]; }
            uint32_t nl = n / 2; int best_axis = -1; double best_cost = INFINITY;
            if (n > 2 && U.depth < kSmallDepth && SR.depth + U.depth < kSahDepth) {
                for (int a = 0; a < 3; a++) {
                    small_sort(L, lane, b, n, a);
                    Box acc; box_empty(acc);
                    for (uint32_t i = n - 1; i > 0; i--) { small_grow(acc, L, lane, L.ord[b + i][lane]); L.right_area[i][lane] = (float)half_area(acc); }
                    box_empty(acc);
                    for (uint32_t i = 1; i < n; i++) {   // i leaves on the left
                        small_grow(acc, L, lane, L.ord[b + i - 1][lane]);
                        const double cost = half_area(acc) * i + (double)L.right_area[i][lane] * (n - i);
                        if (cost < best_cost) { best_cost = cost; best_axis = a; nl = i; }
                    }
                }
                if (best_axis >= 0 && best_axis != 2) small_sort(L, lane, b, n, best_axis);   // the sub-range is in z order now: once more on the winning axis
            }
            const uint32_t nr = n - nl;
            if (nl == 1) child[2 * (size_t)U.k] = ~(int32_t)leaf[L.ord[b][lane]];
            else { child[2 * (size_t)U.k] = (int32_t)(U.k + 1); stack[sp++] = Sub{(uint8_t)b, (uint8_t)nl, (uint8_t)(U.depth + 1), U.k + 1}; }
            if (nr == 1) child[2 * (size_t)U.k + 1] = ~(int32_t)leaf[L.ord[b + nl][lane]];
            else { child[2 * (size_t)U.k + 1] = (int32_t)(U.k + nl); stack[sp++] = Sub{(uint8_t)(b + nl), (uint8_t)nr, (uint8_t)(U.depth + 1), U.k + nl}; }
        }
    }
}
#undef SBOX

__global__ void k_sah_noop() {}
} // namespace
void sah_prewarm(hipStream_t s) { k_sah_noop<<<1, 1, 0, s>>>(); }

hipError_t sah_build_device(Lbvh &l, uint32_t T, hipStream_t s) {
    if (T < 3) return hipSuccess;
    const uint32_t NI = T - 1, max_ranges = T / (kSmall + 1) + 2, max_small = T / 2 + 2;   // a range in the level loop has more than kSmall leaves; a small one at least two
    uint32_t *idx[2] = {nullptr, nullptr}, *range_of[2] = {nullptr, nullptr}, *bins = nullptr, *flags = nullptr, *scan = nullptr, *n_cnt = nullptr;
    Range *ranges[2] = {nullptr, nullptr}; Split *splits = nullptr; SmallRange *small = nullptr; void *tmp = nullptr; size_t tmp_bytes = 0;
    const bool log = (l.log & 1u) != 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto t0 = now(); auto t1 = t0, t2 = t0; uint32_t levels = 0, widest = 0, n_small_host = 0; int idx_cur = 0;   // idx_cur: which of idx[] / range_of[] holds the leaves (a scatter flips it)
    Arena own; Arena &A = l.arena ? *l.arena : own;
    const size_t n_slots = (size_t)T / kMid + 2;
    auto body = [&]() -> hipError_t {
        HIPQ(rocprim::exclusive_scan(nullptr, tmp_bytes, flags, scan, 0u, T, rocprim::plus<uint32_t>(), s));
        HIPQ(A.reserve(6 * Arena::pad((size_t)T * 4) + 2 * Arena::pad((size_t)max_ranges * sizeof(Range)) + Arena::pad(n_slots * kAxes * kBins * 7 * 4) + Arena::pad((size_t)max_ranges * sizeof(Split)) +
                       Arena::pad((size_t)max_small * sizeof(SmallRange)) + Arena::pad(64) + Arena::pad(tmp_bytes ? tmp_bytes : 16)));
        for (int k = 0; k < 2; k++) { idx[k] = A.take<uint32_t>(T); range_of[k] = A.take<uint32_t>(T); ranges[k] = A.take<Range>(max_ranges); }
        bins = A.take<uint32_t>(n_slots * kAxes * kBins * 7); splits = A.take<Split>(max_ranges); small = A.take<SmallRange>(max_small);
        flags = A.take<uint32_t>(T); scan = A.take<uint32_t>(T); n_cnt = A.take<uint32_t>(16);   // n_cnt[0 / 1]: open ranges of this / the next level, [2]: small ranges
        tmp = A.take<char>(tmp_bytes ? tmp_bytes : 16);
        HIPQ(lbvh_claim_trav(l, NI));
        HIPQ(hipStreamSynchronize(s)); t1 = now();
        const uint32_t gT = (T + kBlockB - 1) / kBlockB;
        auto blocks = [](size_t items, uint32_t per_block, uint32_t cap) { size_t b = (items + per_block - 1) / per_block; return (uint32_t)(b < 1 ? 1 : (b > cap ? cap : b)); };
        const uint32_t init[16] = {1u, 0u, 0u, 0u, T > kMid ? 1u : 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // [0 / 1] open ranges of this / the next level, [2] small ranges, [4 / 5] ranges of more than kMid leaves
        HIPQ(hipMemcpy(n_cnt, init, sizeof(init), hipMemcpyHostToDevice));   // (init is a local: a synchronous copy; the stream was synchronised above)
        const bool root_small = T <= kSmall;   // the whole scene is a small range
        k_iota<<<gT, kBlockB, 0, s>>>(T, idx[0], range_of[0], root_small ? kNone : 0u);
        if (root_small) {   // (synchronous copies: the sources are locals of this block -- an asynchronous copy of pageable memory happens to be staged before it returns, but nothing says so)
            const SmallRange root{0, T, 0, 0}; const uint32_t one_zero[3] = {0u, 0u, 1u};
            HIPQ(hipStreamSynchronize(s));
            HIPQ(hipMemcpy(small, &root, sizeof(root), hipMemcpyHostToDevice));
            HIPQ(hipMemcpy(n_cnt, one_zero, sizeof(one_zero), hipMemcpyHostToDevice));
        } else {
            k_root_range<<<1, 1, 0, s>>>(T, l.cbounds, ranges[0]);   // the root's domain: the bounds of the centroids, which the Morton keys were made over
        }
        // The number of open ranges lives on the device (n_cnt[cur]: this level's, n_cnt[cur ^ 1]: the one k_choose counts up for the next); the range-indexed kernels
        // read it there, and the levels are launched four at a time without the host looking -- they are launch- and fence-bound (a dozen 5-50 us kernels between two
        // fences) -- the host asks only every fourth level whether anything is still open; levels behind the last one find no open range and do nothing.
        uint32_t n = root_small ? 0u : 1u, n_big = T > kMid ? 1u : 0u; int cur = 0;   // n_cnt[4 + parity]: how many of a level's ranges have more than kMid leaves
        for (uint32_t level = 0; n > 0 && level < 4096; level++) {
            k_level_reset<<<1, 1, 0, s>>>(n_cnt, cur ^ 1);
            const uint32_t n_up = (uint32_t)std::min<uint64_t>(1ull << std::min(level, 31u), max_ranges);   // at most this many ranges are open (a level doubles them at most)
            if (n_big == 0) {   // every open range fits a block: the level is one launch, the leaves stay in idx[cur]
                const LevelOut lo_{nullptr, ranges[cur ^ 1], n_cnt + (cur ^ 1), small, n_cnt + 2, l.trav_child, l.trav_lo, l.trav_hi, n_cnt + 4 + (cur ^ 1)};
                k_mid<true><<<blocks(n_up, n_up <= 512 ? 1 : (n_up <= 4096 ? 2 : (n_up <= 16384 ? 4 : kBatch)), 32768), kBlockB, 0, s>>>(n_cnt + cur, ranges[cur], idx[idx_cur], l.leaf_lo, l.leaf_hi, lo_);
            } else {
                const LevelOut lo_{splits, ranges[cur ^ 1], n_cnt + (cur ^ 1), small, n_cnt + 2, l.trav_child, l.trav_lo, l.trav_hi, n_cnt + 4 + (cur ^ 1)};
                k_init_level<<<blocks(n_up, 1, 4096), kBlockB, 0, s>>>(n_cnt + cur, ranges[cur], bins);
                k_bin<<<(T + kWin - 1) / kWin, kBlockB, 0, s>>>(T, idx[idx_cur], range_of[idx_cur], l.leaf_lo, l.leaf_hi, ranges[cur], bins);
                k_choose<<<blocks(n_up, 1, 4096), 64, 0, s>>>(n_cnt + cur, ranges[cur], bins, lo_);
                k_mid<false><<<blocks(n_up, n_up <= 512 ? 1 : (n_up <= 4096 ? 2 : (n_up <= 16384 ? 4 : kBatch)), 32768), kBlockB, 0, s>>>(n_cnt + cur, ranges[cur], idx[idx_cur], l.leaf_lo, l.leaf_hi, lo_);
                k_flags<<<gT, kBlockB, 0, s>>>(T, idx[idx_cur], range_of[idx_cur], l.leaf_lo, l.leaf_hi, ranges[cur], splits, flags);
                size_t tb = tmp_bytes;
                HIPQ(rocprim::exclusive_scan(tmp, tb, flags, scan, 0u, T, rocprim::plus<uint32_t>(), s));
                k_scatter<<<gT, kBlockB, 0, s>>>(T, idx[idx_cur], range_of[idx_cur], ranges[cur], splits, flags, scan, idx[idx_cur ^ 1], range_of[idx_cur ^ 1]);
                k_leaf_refs<<<blocks(n_up, 256, 2048), 256, 0, s>>>(n_cnt + cur, ranges[cur], splits, idx[idx_cur ^ 1], l.trav_child);
                idx_cur ^= 1;
            }
            cur ^= 1; levels++;
            if ((level & 3u) == 3u) {
                uint32_t nn[6] = {0, 0, 0, 0, 0, 0};
                HIPQ(hipMemcpyAsync(nn, n_cnt, sizeof(nn), hipMemcpyDeviceToHost, s));
                HIPQ(hipStreamSynchronize(s));
                if (nn[cur] > max_ranges) return hipErrorUnknown;
                n = nn[cur]; widest = n > widest ? n : widest;
                if (n_big) n_big = nn[4 + cur];   // (once zero it stays zero: a child is no larger than its parent)
            }
        }
        if (n != 0) return hipErrorUnknown;
        // everything the levels left: one thread per small range (idx[cur] holds the leaves of every range in its final interval)
        k_small<<<blocks(max_small, 64, 16384), 64, 0, s>>>(n_cnt + 2, small, idx[idx_cur], l.leaf_lo, l.leaf_hi, l.trav_child, l.trav_lo, l.trav_hi);
        if (log) { HIPQ(hipMemcpyAsync(&n_small_host, n_cnt + 2, 4, hipMemcpyDeviceToHost, s)); HIPQ(hipStreamSynchronize(s)); }
        t2 = now();
        HIPQ(hipGetLastError());
        launch_emit_nodes(l, T, s);
        HIPQ(hipGetLastError());
        HIPQ(hipStreamSynchronize(s));
        return hipSuccess;
    };
    hipError_t err = body();
    auto t3 = now();
    own.release();
    if (log) std::fprintf(stderr, "[art] sah_build_device %u leaves: wait + allocations %.1f ms, %u levels launched (at most %u open ranges seen at a fence), %u small ranges %.1f, node records %.1f, frees %.1f\n", T, ms(t0, t1), levels, widest, n_small_host, ms(t1, t2), ms(t2, t3), ms(t3, now()));
    return err;
}

} // namespace art
