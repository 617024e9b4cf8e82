// The PREFER_FAST_TRACE traversal tree built on the device: the same binned surface-area heuristic as art_sah.hip (32 bins on each of the
// three axes, binned by box centroid, pre-order node numbering: the left subtree follows its parent, the right one starts nl nodes on),
// level by level over ALL open ranges at once.  Every node box is an exact min/max union of leaf boxes (unions go through an
// order-preserving float <-> uint key, so the atomics are integer min / max), hence frames cannot depend on which builder ran.
//
// One level = a handful of launches over the T leaf positions:
//   k_centroid_bounds   per position -> atomic min/max into its range's centroid box
//   k_bin               per position -> its bin on each axis: leaf box + count (21 atomics; the top levels contend on 672 words, ~0.4 ms each)
//   k_choose            per range    -> node box, best (axis, bin) by SAH or the median past the depth guard; opens the child ranges
//   k_flags + scan + k_scatter       -> stable partition of every range at once (one exclusive scan over T flags)
//   k_leaf_refs         per range    -> child references of one-leaf sides (known only after the partition)
// and one 4-byte read-back (how many ranges the next level has).
#include "art_internal.h"
#include <rocprim/rocprim.hpp>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace art {
namespace {

#define HIPQ(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

constexpr int kBins = 32;
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kSahDepth = 48; // past it ranges are halved: the tree stays within kSahDepth + log2(T) levels (the walks' stacks)
constexpr uint32_t kBlockB = 256;

struct Range { uint32_t b, e, k, depth; };               // leaves idx[b, e) -> internal node k
struct Split { uint32_t axis, bin, nl, left, right; };   // axis 3: by position (median); left / right: the child ranges' ids in the next level, or kNone

__device__ __forceinline__ uint32_t fkey(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float fkey_inv(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }
__device__ __forceinline__ float centroid(const float *lo, const float *hi, uint32_t leaf, int a) { return 0.5f * lo[3 * (size_t)leaf + a] + 0.5f * hi[3 * (size_t)leaf + a]; }
__device__ __forceinline__ int bin_of(float c, float c0, float sc) { int b = (int)((c - c0) * sc); return b < 0 ? 0 : (b >= kBins ? kBins - 1 : b); }

// cb: [range][6] keys (lo xyz initialised to ~0, hi xyz to 0); bins: [range][axis][bin][7] = lo xyz keys, hi xyz keys, count
// n_cb ranges' centroid boxes (0: leave them) and the bins of a window of n_win ranges
// (the number of open ranges of a level is read from the device -- n_dev, the counter the level before filled -- so that a level can be launched before the
// host knows it: grid-stride loops over whatever the launch was given)
__global__ void k_init_level(const uint32_t *__restrict__ n_dev, bool with_cb, uint32_t r0, uint32_t win_cap, uint32_t *cb, uint32_t *bins) {
    const uint32_t n = *n_dev;
    const uint32_t n_win = n > r0 ? (n - r0 < win_cap ? n - r0 : win_cap) : 0u;
    const size_t n_cb_words = with_cb ? (size_t)n * 6 : 0, n_bin_words = (size_t)n_win * 3 * kBins * 7, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cb_words || i < n_bin_words; i += stride) {
        if (i < n_cb_words) cb[i] = (i % 6) < 3 ? 0xFFFFFFFFu : 0u;
        if (i < n_bin_words) { uint32_t w = (uint32_t)(i % 7); bins[i] = w < 3 ? 0xFFFFFFFFu : 0u; }
    }
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) { for (int m = 32; m > 0; m >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, m)); return v; }
// Near the root thousands of leaves share a range: the centroid bounds are reduced per wave (segmented), the bins per block in LDS when the block has ONE range
// -- 262 k leaves on the root's six words took 3.3 ms of serialised atomics otherwise.
__global__ __launch_bounds__(kBlockB) void k_centroid_bounds(uint32_t T, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ range_of, const float *__restrict__ lo, const float *__restrict__ hi, uint32_t *cb) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t r = i < T ? range_of[i] : kNone;
    const bool valid = r != kNone;
    uint32_t klo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, khi[3] = {0u, 0u, 0u};
    if (valid) { uint32_t leaf = idx[i]; for (int a = 0; a < 3; a++) klo[a] = khi[a] = fkey(centroid(lo, hi, leaf, a)); }
    if (wave_min(valid ? r : 0xFFFFFFFFu) == 0xFFFFFFFFu) return; // no leaf of an open range in this wave
    // A range is an interval of positions, so the lanes of one range are neighbours: a segmented scan over the wave leaves each range's
    // bounds in its last lane, and only that lane sends atomics (up to 64 times fewer; all of them near the root).
    const uint32_t lane = threadIdx.x & 63u;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t ro = (uint32_t)__shfl_up((int)r, off);
        uint32_t ol[3], oh[3];
        for (int a = 0; a < 3; a++) { ol[a] = (uint32_t)__shfl_up((int)klo[a], off); oh[a] = (uint32_t)__shfl_up((int)khi[a], off); }
        if (lane >= (uint32_t)off && ro == r) for (int a = 0; a < 3; a++) { klo[a] = min(klo[a], ol[a]); khi[a] = max(khi[a], oh[a]); }
    }
    const uint32_t rn = (uint32_t)__shfl_down((int)r, 1);
    if (valid && (lane == 63u || rn != r)) for (int a = 0; a < 3; a++) { atomicMin(&cb[(size_t)r * 6 + a], klo[a]); atomicMax(&cb[(size_t)r * 6 + 3 + a], khi[a]); }
}
__global__ __launch_bounds__(kBlockB) void k_bin(uint32_t T, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ range_of, const float *__restrict__ lo, const float *__restrict__ hi,
                                                 const uint32_t *__restrict__ cb, uint32_t *bins, uint32_t r0, uint32_t n_win /* at most: the window's capacity */) {
    // bins holds the ranges [r0, r0 + n_win) of this level (a window: the bins of ALL ranges of a deep level would be 1344 bytes per triangle)
    __shared__ uint32_t s_lo, s_hi, s_bins[3 * kBins * 7];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t r = i < T ? range_of[i] : kNone;
    const bool valid = r != kNone && r >= r0 && r - r0 < n_win;
    if (threadIdx.x == 0) { s_lo = 0xFFFFFFFFu; s_hi = 0u; }
    for (uint32_t w = threadIdx.x; w < 3 * kBins * 7; w += kBlockB) s_bins[w] = (w % 7) < 3 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    if (valid) { atomicMin(&s_lo, r); atomicMax(&s_hi, r); }
    __syncthreads();
    if (s_lo == 0xFFFFFFFFu) return;                      // nothing open in this block
    const bool one = s_lo == s_hi;                        // the whole block bins into ONE range: histogram in LDS, then one global atomic per touched word
    uint32_t *base = one ? s_bins : bins + (size_t)(r - r0) * 3 * kBins * 7;
    if (valid) {
        uint32_t leaf = idx[i];
        uint32_t kl[3], kh[3];
        for (int a = 0; a < 3; a++) { kl[a] = fkey(lo[3 * (size_t)leaf + a]); kh[a] = fkey(hi[3 * (size_t)leaf + a]); }
        for (int a = 0; a < 3; a++) {
            float c0 = fkey_inv(cb[(size_t)r * 6 + a]), c1 = fkey_inv(cb[(size_t)r * 6 + 3 + a]), ext = c1 - c0;
            int b = ext > 0.0f ? bin_of(centroid(lo, hi, leaf, a), c0, (float)kBins / ext) : 0; // a flat axis: everything in bin 0, never chosen
            uint32_t *w = base + ((size_t)a * kBins + b) * 7;
            for (int k = 0; k < 3; k++) { atomicMin(&w[k], kl[k]); atomicMax(&w[3 + k], kh[k]); }
            atomicAdd(&w[6], 1u);
        }
    }
    if (!one) return;
    __syncthreads();
    uint32_t *g = bins + (size_t)(s_lo - r0) * 3 * kBins * 7;
    for (uint32_t w = threadIdx.x; w < 3 * kBins * 7; w += kBlockB) {
        const uint32_t k = w % 7, v = s_bins[w];
        if (s_bins[w - k + 6] == 0) continue;             // empty bin
        if (k < 3) atomicMin(&g[w], v); else if (k < 6) atomicMax(&g[w], v); else atomicAdd(&g[w], v);
    }
}
struct Box { float lo[3], hi[3]; };
__device__ __forceinline__ void box_empty(Box &b) { for (int k = 0; k < 3; k++) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; } }
__device__ __forceinline__ void box_grow(Box &b, const uint32_t *w) { for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], fkey_inv(w[k])); b.hi[k] = fmaxf(b.hi[k], fkey_inv(w[3 + k])); } }
__device__ __forceinline__ double half_area(const Box &b) { double dx = (double)b.hi[0] - b.lo[0], dy = (double)b.hi[1] - b.lo[1], dz = (double)b.hi[2] - b.lo[2]; return dx < 0 ? 0.0 : dx * dy + dy * dz + dz * dx; }

__global__ void k_choose(const uint32_t *__restrict__ n_dev, uint32_t win_cap, const Range *__restrict__ ranges, const uint32_t *__restrict__ cb, const uint32_t *__restrict__ bins, Split *splits, Range *next, uint32_t *n_next,
                         int32_t *child, float *nlo, float *nhi, uint32_t r0) {
    const uint32_t n_all = *n_dev, n_ranges = n_all > r0 ? (n_all - r0 < win_cap ? n_all : r0 + win_cap) : r0;   // end of the window [r0, n_ranges) whose bins are in `bins`
    bins -= (size_t)r0 * 3 * kBins * 7;                        // indexed by r below
    for (uint32_t r = r0 + blockIdx.x * blockDim.x + threadIdx.x; r < n_ranges; r += gridDim.x * blockDim.x) {
    const Range R = ranges[r];
    const uint32_t n = R.e - R.b;
    Box node; box_empty(node);
    for (int i = 0; i < kBins; i++) { const uint32_t *w = bins + (((size_t)r * 3 + 0) * kBins + i) * 7; if (w[6]) box_grow(node, w); }
    for (int k = 0; k < 3; k++) { nlo[3 * (size_t)R.k + k] = node.lo[k]; nhi[3 * (size_t)R.k + k] = node.hi[k]; }
    int best_axis = -1, best_bin = 0; uint32_t best_left = 0; double best_cost = INFINITY;
    if (n > 2 && R.depth < kSahDepth) {
        for (int a = 0; a < 3; a++) {
            if (!(fkey_inv(cb[(size_t)r * 6 + 3 + a]) > fkey_inv(cb[(size_t)r * 6 + a]))) continue;
            const uint32_t *wa = bins + ((size_t)r * 3 + a) * kBins * 7;
            double right_area[kBins]; uint32_t right_cnt[kBins];
            Box acc; box_empty(acc); uint32_t c = 0;
            for (int i = kBins - 1; i > 0; i--) { if (wa[i * 7 + 6]) box_grow(acc, wa + i * 7); c += wa[i * 7 + 6]; right_area[i] = half_area(acc); right_cnt[i] = c; }
            box_empty(acc); c = 0;
            for (int i = 0; i < kBins - 1; i++) {
                if (wa[i * 7 + 6]) box_grow(acc, wa + i * 7);
                c += wa[i * 7 + 6];
                if (c == 0 || right_cnt[i + 1] == 0) continue;
                double cost = half_area(acc) * c + right_area[i + 1] * right_cnt[i + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = i; best_left = c; }
            }
        }
    }
    Split S;
    if (best_axis < 0) { S.axis = 3; S.bin = 0; S.nl = n / 2; }       // two leaves, coincident centroids, or past the depth guard
    else { S.axis = (uint32_t)best_axis; S.bin = (uint32_t)best_bin; S.nl = best_left; }
    const uint32_t nl = S.nl, nr = n - nl;
    S.left = kNone; S.right = kNone;
    if (nl > 1) { S.left = atomicAdd(n_next, 1u); next[S.left] = Range{R.b, R.b + nl, R.k + 1, R.depth + 1}; child[2 * (size_t)R.k] = (int32_t)(R.k + 1); }
    if (nr > 1) { S.right = atomicAdd(n_next, 1u); next[S.right] = Range{R.b + nl, R.e, R.k + nl, R.depth + 1}; child[2 * (size_t)R.k + 1] = (int32_t)(R.k + nl); }
    splits[r] = S;
    }
}
__global__ __launch_bounds__(kBlockB) void k_flags(uint32_t T, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ range_of, const float *__restrict__ lo, const float *__restrict__ hi,
                                                   const Range *__restrict__ ranges, const Split *__restrict__ splits, const uint32_t *__restrict__ cb, uint32_t *flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    uint32_t r = range_of[i], f = 0;
    if (r != kNone) {
        const Split S = splits[r];
        if (S.axis == 3) f = (i - ranges[r].b) < S.nl;
        else {
            float c0 = fkey_inv(cb[(size_t)r * 6 + S.axis]), c1 = fkey_inv(cb[(size_t)r * 6 + 3 + S.axis]);
            f = bin_of(centroid(lo, hi, idx[i], (int)S.axis), c0, (float)kBins / (c1 - c0)) <= (int)S.bin;
        }
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
    range_of2[dst] = flags[i] ? S.left : S.right;
}
__global__ void k_leaf_refs(const uint32_t *__restrict__ n_dev, const Range *__restrict__ ranges, const Split *__restrict__ splits, const uint32_t *__restrict__ idx2, int32_t *child) {
    const uint32_t n_ranges = *n_dev;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_ranges; r += gridDim.x * blockDim.x) {
        const Range R = ranges[r]; const Split S = splits[r];
        if (S.nl == 1) child[2 * (size_t)R.k] = ~(int32_t)idx2[R.b];
        if (R.e - R.b - S.nl == 1) child[2 * (size_t)R.k + 1] = ~(int32_t)idx2[R.b + S.nl];
    }
}
__global__ void k_iota(uint32_t T, uint32_t *idx, uint32_t *range_of) { uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < T) { idx[i] = i; range_of[i] = 0; } }

} // namespace

hipError_t sah_build_device(Lbvh &l, uint32_t T, hipStream_t s) {
    if (T < 3) return hipSuccess;
    const uint32_t NI = T - 1, max_ranges = T / 2 + 1;
    // Bins are 3 axes x 32 bins x 7 words = 2688 bytes per open range.  A deep level of a big scene has ~T/4 open ranges (3.8 GB for the
    // 2.8 M triangles of config 4 if all had bins at once), so a level is binned and split in windows of at most kWindow ranges: 352 MB
    // whatever T is; config 2 (T/2 = 131 k ranges at most) still takes one pass per level, config 4's deepest levels take up to 11.
    constexpr uint32_t kWindow = 1u << 17;
    const uint32_t win_cap = max_ranges < kWindow ? max_ranges : kWindow;
    uint32_t *idx[2] = {nullptr, nullptr}, *range_of[2] = {nullptr, nullptr}, *cb = nullptr, *bins = nullptr, *flags = nullptr, *scan = nullptr, *n_cnt = nullptr;
    Range *ranges[2] = {nullptr, nullptr}; Split *splits = nullptr; void *tmp = nullptr; size_t tmp_bytes = 0;
    const bool log = (l.log & 1u) != 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto t0 = now(); auto t1 = t0, t2 = t0; uint32_t levels = 0, widest = 0;
    auto body = [&]() -> hipError_t {
        for (int k = 0; k < 2; k++) { HIPQ(hipMalloc(&idx[k], (size_t)T * 4)); HIPQ(hipMalloc(&range_of[k], (size_t)T * 4)); HIPQ(hipMalloc(&ranges[k], (size_t)max_ranges * sizeof(Range))); }
        HIPQ(hipMalloc(&cb, (size_t)max_ranges * 6 * 4)); HIPQ(hipMalloc(&bins, (size_t)win_cap * 3 * kBins * 7 * 4)); HIPQ(hipMalloc(&splits, (size_t)max_ranges * sizeof(Split)));
        HIPQ(hipMalloc(&flags, (size_t)T * 4)); HIPQ(hipMalloc(&scan, (size_t)T * 4)); HIPQ(hipMalloc(&n_cnt, 8));
        if (!l.trav_child) { HIPQ(hipMalloc(&l.trav_child, (size_t)NI * 8)); HIPQ(hipMalloc(&l.trav_lo, (size_t)NI * 12)); HIPQ(hipMalloc(&l.trav_hi, (size_t)NI * 12)); }
        HIPQ(rocprim::exclusive_scan(nullptr, tmp_bytes, flags, scan, 0u, T, rocprim::plus<uint32_t>(), s));
        HIPQ(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
        HIPQ(hipStreamSynchronize(s)); t1 = now();
        const uint32_t gT = (T + kBlockB - 1) / kBlockB;
        k_iota<<<gT, kBlockB, 0, s>>>(T, idx[0], range_of[0]);
        Range root{0, T, 0, 0};
        HIPQ(hipMemcpyAsync(ranges[0], &root, sizeof(root), hipMemcpyHostToDevice, s));
        // The number of open ranges lives on the device (n_cnt[cur]: this level's, n_cnt[cur ^ 1]: the one k_choose counts up for the next); the range-indexed kernels
        // read it there.  A scene whose levels fit two windows of bins (T <= 4 * kWindow: config 2) is launched four levels at a time without the host looking -- its
        // levels are launch- and fence-bound (23 levels of 262 k leaves: 7.3 ms with a fence per level, a dozen 5 us kernels between two fences) -- and the host asks
        // only every fourth level whether anything is still open; levels behind the last one find no open range and do nothing.  A bigger scene needs the count for its
        // windows, and its levels are work-bound anyway: one fence per level as before.
        const bool pipelined = max_ranges <= 2 * (uint64_t)win_cap;   // (config 2: T / 2 + 1 = 131 409 possible ranges against a window of 131 072 -- its deepest levels launch a second, nearly empty window)
        const uint32_t one = 1;
        HIPQ(hipMemcpyAsync(n_cnt, &one, 4, hipMemcpyHostToDevice, s));
        uint32_t n = 1; int cur = 0;
        auto blocks = [](size_t items, uint32_t per_block, uint32_t cap) { size_t b = (items + per_block - 1) / per_block; return (uint32_t)(b < 1 ? 1 : (b > cap ? cap : b)); };
        for (uint32_t level = 0; n > 0 && level < 4096; level++) {
            HIPQ(hipMemsetAsync(n_cnt + (cur ^ 1), 0, 4, s));
            const uint32_t n_up = pipelined ? (uint32_t)std::min<uint64_t>(1ull << std::min(level, 31u), max_ranges) : n;   // at most this many ranges are open (a level doubles them at most)
            for (uint32_t r0 = 0; r0 < n_up; r0 += win_cap) {
                const uint32_t nw = n_up - r0 < win_cap ? n_up - r0 : win_cap;
                const size_t words = std::max((size_t)nw * 3 * kBins * 7, r0 == 0 ? (size_t)n_up * 6 : (size_t)0);   // the centroid boxes of the whole level are cleared with its first window
                k_init_level<<<blocks(words, kBlockB, 8192), kBlockB, 0, s>>>(n_cnt + cur, r0 == 0, r0, win_cap, cb, bins);
                if (r0 == 0) k_centroid_bounds<<<gT, kBlockB, 0, s>>>(T, idx[cur], range_of[cur], l.leaf_lo, l.leaf_hi, cb);
                k_bin<<<gT, kBlockB, 0, s>>>(T, idx[cur], range_of[cur], l.leaf_lo, l.leaf_hi, cb, bins, r0, win_cap);
                k_choose<<<blocks(nw, 64, 4096), 64, 0, s>>>(n_cnt + cur, win_cap, ranges[cur], cb, bins, splits, ranges[cur ^ 1], n_cnt + (cur ^ 1), l.trav_child, l.trav_lo, l.trav_hi, r0);
            }
            k_flags<<<gT, kBlockB, 0, s>>>(T, idx[cur], range_of[cur], l.leaf_lo, l.leaf_hi, ranges[cur], splits, cb, flags);
            size_t tb = tmp_bytes;
            HIPQ(rocprim::exclusive_scan(tmp, tb, flags, scan, 0u, T, rocprim::plus<uint32_t>(), s));
            k_scatter<<<gT, kBlockB, 0, s>>>(T, idx[cur], range_of[cur], ranges[cur], splits, flags, scan, idx[cur ^ 1], range_of[cur ^ 1]);
            k_leaf_refs<<<blocks(n_up, 256, 2048), 256, 0, s>>>(n_cnt + cur, ranges[cur], splits, idx[cur ^ 1], l.trav_child);
            cur ^= 1; levels++;
            if (!pipelined || (level & 3u) == 3u) {
                uint32_t nn = 0;
                HIPQ(hipMemcpyAsync(&nn, n_cnt + cur, 4, hipMemcpyDeviceToHost, s));
                HIPQ(hipStreamSynchronize(s));
                if (nn > max_ranges) return hipErrorUnknown;
                n = nn; widest = n > widest ? n : widest;
            }
        }
        t2 = now();
        if (n != 0) return hipErrorUnknown;
        HIPQ(hipGetLastError());
        launch_emit_nodes(l, T, s);
        HIPQ(hipGetLastError());
        HIPQ(hipStreamSynchronize(s));
        return hipSuccess;
    };
    hipError_t err = body();
    auto t3 = now();
    for (int k = 0; k < 2; k++) { hipFree(idx[k]); hipFree(range_of[k]); hipFree(ranges[k]); }
    hipFree(cb); hipFree(bins); hipFree(splits); hipFree(flags); hipFree(scan); hipFree(n_cnt); hipFree(tmp);
    if (log) std::fprintf(stderr, "[art] sah_build_device %u leaves: wait + allocations %.1f ms, %u levels launched (at most %u open ranges seen) %.1f, node records %.1f, frees %.1f\n", T, ms(t0, t1), levels, widest, ms(t1, t2), ms(t2, t3), ms(t3, now()));
    return err;
}

} // namespace art
