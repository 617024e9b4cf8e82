// art_build.hip -- device LBVH builder over the world-space triangle soup (gfx950).
//
// Replaces the closed driver work behind VkBlasBuilder::build_blas_from_geometry (vk_blas_builder.rs:88-170,
// vkCmdBuildAccelerationStructuresKHR, PREFER_FAST_TRACE) and VkTlasBuilder::recreate_tlas
// (vk_tlas_builder.rs:38-233): instances are folded into the soup (one instance per model, renderer.rs:641-650).
//
// Pipeline: soup (index fetch + object->world) -> centroid bounds (ordered-int atomics) -> Morton keys ->
// stable radix sort (rocPRIM) -> Karras 2012 radix tree -> bottom-up AABB refit (arrival counters) ->
// 64-byte traversal nodes + 64-byte leaf-ordered triangle records; the 4-wide collapse of the traversal tree, level by level (wide_build).
#include "art_internal.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cmath>
#include <vector>

namespace art {

#define HIPQ(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

__device__ inline uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ inline float ord2f(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }
constexpr uint32_t kCbSlots = 64;

__device__ inline float3 xform_point(const float *m, float x, float y, float z) {
    return make_float3(((m[0] * x + m[1] * y) + m[2] * z) + m[3], ((m[4] * x + m[5] * y) + m[6] * z) + m[7],
                       ((m[8] * x + m[9] * y) + m[10] * z) + m[11]);
}

// one thread per triangle: fetch indices + vertices, transform, write world triangle, its AABB, reduce centroid bounds
__global__ __launch_bounds__(256) void k_soup(const DevPrim *__restrict__ prims, uint32_t n_prims, const uint32_t *__restrict__ first_tri,
                                              uint32_t T, float *__restrict__ triw /*T*9*/, float *__restrict__ tlo, float *__restrict__ thi,
                                              uint32_t *__restrict__ tri_prim, uint32_t *__restrict__ cslots /*kCbSlots x 32 words: 6 ordered ints each, a slot per 128-byte line*/) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    float cx = 0, cy = 0, cz = 0;
    bool on = g < T;
    if (on) {
        uint32_t lo = 0, hi = n_prims; // last p with first_tri[p] <= g
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (first_tri[mid] <= g) lo = mid; else hi = mid; }
        const DevPrim &P = prims[lo];
        uint32_t t = g - P.first_tri;
        uint32_t i0, i1, i2;
        if (P.single_index_size == 2) { const uint16_t *ix = (const uint16_t *)P.indices + 3 * (size_t)t; i0 = ix[0]; i1 = ix[1]; i2 = ix[2]; }
        else { const uint32_t *ix = (const uint32_t *)P.indices + 3 * (size_t)t; i0 = ix[0]; i1 = ix[1]; i2 = ix[2]; }
        const float *a = P.vertices + (size_t)i0 * 12, *b = P.vertices + (size_t)i1 * 12, *c = P.vertices + (size_t)i2 * 12;
        float3 w0 = xform_point(P.o2w, a[0], a[1], a[2]), w1 = xform_point(P.o2w, b[0], b[1], b[2]), w2 = xform_point(P.o2w, c[0], c[1], c[2]);
        float *o = triw + (size_t)g * 9;
        o[0] = w0.x; o[1] = w0.y; o[2] = w0.z; o[3] = w1.x; o[4] = w1.y; o[5] = w1.z; o[6] = w2.x; o[7] = w2.y; o[8] = w2.z;
        float lx = fminf(fminf(w0.x, w1.x), w2.x), ly = fminf(fminf(w0.y, w1.y), w2.y), lz = fminf(fminf(w0.z, w1.z), w2.z);
        float hx = fmaxf(fmaxf(w0.x, w1.x), w2.x), hy = fmaxf(fmaxf(w0.y, w1.y), w2.y), hz = fmaxf(fmaxf(w0.z, w1.z), w2.z);
        tlo[3 * (size_t)g] = lx; tlo[3 * (size_t)g + 1] = ly; tlo[3 * (size_t)g + 2] = lz;
        thi[3 * (size_t)g] = hx; thi[3 * (size_t)g + 1] = hy; thi[3 * (size_t)g + 2] = hz;
        tri_prim[g] = lo;
        cx = (lx + hx) * 0.5f; cy = (ly + hy) * 0.5f; cz = (lz + hz) * 0.5f;
    }
    // wave reduction of the ordered keys, then one atomic per wave
    uint32_t kmin[3] = {on ? f2ord(cx) : 0xFFFFFFFFu, on ? f2ord(cy) : 0xFFFFFFFFu, on ? f2ord(cz) : 0xFFFFFFFFu};
    uint32_t kmax[3] = {on ? f2ord(cx) : 0u, on ? f2ord(cy) : 0u, on ? f2ord(cz) : 0u};
    for (int off = 32; off >= 1; off >>= 1)
        for (int k = 0; k < 3; k++) {
            kmin[k] = min(kmin[k], (uint32_t)__shfl_xor((int)kmin[k], off));
            kmax[k] = max(kmax[k], (uint32_t)__shfl_xor((int)kmax[k], off));
        }
    // (every wave of the launch on the same six words was 2.7 of the 3.0 ms this kernel took for the 2.8 M triangles of config 4: the waves are dealt to 64 slots, k_cb_fold joins them)
    if ((threadIdx.x & 63) == 0) {
        uint32_t *cb = cslots + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % kCbSlots) * 32;
        for (int k = 0; k < 3; k++) { atomicMin(&cb[k], kmin[k]); atomicMax(&cb[3 + k], kmax[k]); }
    }
}
__global__ __launch_bounds__(64) void k_cb_init(uint32_t *cslots) { for (int k = 0; k < 6; k++) cslots[threadIdx.x * 32 + k] = k < 3 ? 0xFFFFFFFFu : 0u; }
__global__ __launch_bounds__(64) void k_cb_fold(const uint32_t *__restrict__ cslots, uint32_t *__restrict__ cbounds) {
    uint32_t v[6];
    for (int k = 0; k < 6; k++) v[k] = cslots[threadIdx.x * 32 + k];
    for (int off = 32; off >= 1; off >>= 1) for (int k = 0; k < 6; k++) { uint32_t o = (uint32_t)__shfl_xor((int)v[k], off); v[k] = k < 3 ? min(v[k], o) : max(v[k], o); }
    if (threadIdx.x == 0) for (int k = 0; k < 6; k++) cbounds[k] = v[k];
}

__device__ inline uint32_t expand10(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ inline uint64_t expand21(uint64_t v) {
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull; v = (v | (v << 16)) & 0x1f0000ff0000ffull; v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull; v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ __launch_bounds__(256) void k_morton(uint32_t T, uint32_t bits, const float *__restrict__ tlo, const float *__restrict__ thi,
                                                const uint32_t *__restrict__ cbounds, uint64_t *__restrict__ keys, uint32_t *__restrict__ gids) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= T) return;
    float cmin[3], cmax[3];
    for (int k = 0; k < 3; k++) { cmin[k] = ord2f(cbounds[k]); cmax[k] = ord2f(cbounds[3 + k]); }
    float cells = bits == 30 ? 1024.0f : 2097152.0f;
    float q[3];
    for (int k = 0; k < 3; k++) {
        float ext = cmax[k] - cmin[k];
        float sc = ext > 0.0f ? cells / ext : 0.0f;
        float c = (tlo[3 * (size_t)g + k] + thi[3 * (size_t)g + k]) * 0.5f;
        q[k] = fminf(fmaxf((c - cmin[k]) * sc, 0.0f), cells - 1.0f);
    }
    uint64_t key;
    if (bits == 30) key = ((uint64_t)expand10((uint32_t)q[0]) << 2) | ((uint64_t)expand10((uint32_t)q[1]) << 1) | (uint64_t)expand10((uint32_t)q[2]);
    else key = (expand21((uint64_t)q[0]) << 2) | (expand21((uint64_t)q[1]) << 1) | expand21((uint64_t)q[2]);
    keys[g] = key; gids[g] = g;
}

__device__ inline int delta_fn(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ gid, int T, int i, int j) {
    if (j < 0 || j >= T) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)(gid[i] ^ gid[j]));
}

// Karras 2012, one thread per internal node
__global__ __launch_bounds__(256) void k_karras(int T, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ gid, int32_t *__restrict__ child,
                                                int32_t *__restrict__ parent_int, int32_t *__restrict__ parent_leaf) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T - 1) return;
    if (i == 0) parent_int[0] = -1;
    int d = delta_fn(keys, gid, T, i, i + 1) - delta_fn(keys, gid, T, i, i - 1) >= 0 ? 1 : -1;
    int dmin = delta_fn(keys, gid, T, i, i - d);
    int lmax = 2;
    while (delta_fn(keys, gid, T, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta_fn(keys, gid, T, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta_fn(keys, gid, T, i, j);
    int sp = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (delta_fn(keys, gid, T, i, i + (sp + t) * d) > dnode) sp += t;
    } while (t > 1);
    int gamma = i + sp * d + (d < 0 ? -1 : 0);
    int lo = min(i, j), hi = max(i, j);
    if (lo == gamma) { child[2 * i] = ~gamma; parent_leaf[gamma] = i; } else { child[2 * i] = gamma; parent_int[gamma] = i; }
    if (hi == gamma + 1) { child[2 * i + 1] = ~(gamma + 1); parent_leaf[gamma + 1] = i; } else { child[2 * i + 1] = gamma + 1; parent_int[gamma + 1] = i; }
}

// leaf boxes + leaf-ordered triangle records
__global__ __launch_bounds__(256) void k_leaves(uint32_t T, const uint32_t *__restrict__ leaf_gid, const float *__restrict__ triw, const float *__restrict__ tlo,
                                                const float *__restrict__ thi, const uint32_t *__restrict__ tri_prim, const uint32_t *__restrict__ first_tri,
                                                float *__restrict__ leaf_lo, float *__restrict__ leaf_hi, DevTri *__restrict__ tris,
                                                const DevPrim *__restrict__ prims, DevShadeTri *__restrict__ shade_tris) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= T) return;
    uint32_t g = leaf_gid[p];
    for (int k = 0; k < 3; k++) { leaf_lo[3 * (size_t)p + k] = tlo[3 * (size_t)g + k]; leaf_hi[3 * (size_t)p + k] = thi[3 * (size_t)g + k]; }
    const float *w = triw + (size_t)g * 9;
    uint32_t pr = tri_prim[g];
    DevTri t;
    for (int k = 0; k < 3; k++) {
        t.f[k] = w[k]; t.f[3 + k] = w[3 + k] - w[k]; t.f[6 + k] = w[6 + k] - w[k];
        t.f[9 + k] = fminf(fminf(w[k], w[3 + k]), w[6 + k]); t.f[12 + k] = fmaxf(fmaxf(w[k], w[3 + k]), w[6 + k]);
    }
    t.f[15] = __uint_as_float(g);
    tris[p] = t;
    // shading record: get_indices + three vertex fetches of raytrace.rgen.glsl:107-114, done once
    const DevPrim &P = prims[pr];
    uint32_t tl = g - first_tri[pr], ix[3];
    if (P.single_index_size == 2) { const uint16_t *q = (const uint16_t *)P.indices + 3 * (size_t)tl; ix[0] = q[0]; ix[1] = q[1]; ix[2] = q[2]; }
    else { const uint32_t *q = (const uint32_t *)P.indices + 3 * (size_t)tl; ix[0] = q[0]; ix[1] = q[1]; ix[2] = q[2]; }
    DevShadeTri st;
    for (int k = 0; k < 3; k++) {
        const float *v = P.vertices + (size_t)ix[k] * 12;
        for (int j = 0; j < 3; j++) { st.f[3 * k + j] = v[j]; st.f[15 + 3 * k + j] = v[5 + j]; st.f[24 + 3 * k + j] = v[8 + j]; }
        st.f[9 + 2 * k] = v[3]; st.f[10 + 2 * k] = v[4];
        if (k == 0) st.f[33] = v[11];
    }
    st.f[34] = __uint_as_float(pr); st.f[35] = 0.f;
    shade_tris[p] = st;
}

// bottom-up refit: the second thread to arrive at a node owns it (its sibling's box is complete and visible)
__global__ __launch_bounds__(256) void k_refit(uint32_t T, const int32_t *__restrict__ child, const int32_t *__restrict__ parent_int,
                                               const int32_t *__restrict__ parent_leaf, const float *__restrict__ leaf_lo, const float *__restrict__ leaf_hi,
                                               float *node_lo, float *node_hi, uint32_t *arrive) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= T) return;
    int n = parent_leaf[p];
    while (n >= 0) {
        __threadfence();                       // release: this thread's boxes below n are written back
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // keep the write-back ahead of the arrival (hipcc may drop the wait)
        if (atomicAdd(&arrive[n], 1u) == 0u) return;
        __threadfence();                       // acquire: see the sibling subtree's boxes
        float l[2][3], h[2][3];
        for (int c = 0; c < 2; c++) {
            int ch = child[2 * n + c];
            const float *cl = ch < 0 ? leaf_lo + 3 * (size_t)(~ch) : node_lo + 3 * (size_t)ch;
            const float *chh = ch < 0 ? leaf_hi + 3 * (size_t)(~ch) : node_hi + 3 * (size_t)ch;
            for (int k = 0; k < 3; k++) { l[c][k] = __builtin_nontemporal_load(cl + k); h[c][k] = __builtin_nontemporal_load(chh + k); }
        }
        // (a masked subtree -- a refit after art_scene_set_primitive_enabled -- is "nowhere" and leaves the union alone; a build never sees one)
        const bool n0 = box_nowhere(l[0][0]), n1 = box_nowhere(l[1][0]);
        for (int k = 0; k < 3; k++) {
            node_lo[3 * (size_t)n + k] = n0 ? l[1][k] : (n1 ? l[0][k] : fminf(l[0][k], l[1][k]));
            node_hi[3 * (size_t)n + k] = n0 ? h[1][k] : (n1 ? h[0][k] : fmaxf(h[0][k], h[1][k]));
        }
        n = parent_int[n];
    }
}

__global__ __launch_bounds__(256) void k_emit_nodes(uint32_t T, const int32_t *__restrict__ child, const float *__restrict__ node_lo, const float *__restrict__ node_hi,
                                                    const float *__restrict__ leaf_lo, const float *__restrict__ leaf_hi, DevNode *__restrict__ nodes) {
    uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (T == 1) { // single triangle: a root whose second child can never be hit
        if (n == 0) {
            DevNode d;
            d.q[0] = make_float4(leaf_lo[0], leaf_lo[1], leaf_lo[2], leaf_hi[0]);
            d.q[1] = make_float4(leaf_hi[1], leaf_hi[2], 3.0e38f, 3.0e38f); // a far-away point box: every slab test fails
            d.q[2] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f);
            d.q[3] = make_float4(__int_as_float(~0), __int_as_float(~0), 0.f, 0.f);
            nodes[0] = d;
        }
        return;
    }
    if (n >= T - 1) return;
    int c0 = child[2 * n], c1 = child[2 * n + 1];
    const float *l0 = c0 < 0 ? leaf_lo + 3 * (size_t)(~c0) : node_lo + 3 * (size_t)c0, *h0 = c0 < 0 ? leaf_hi + 3 * (size_t)(~c0) : node_hi + 3 * (size_t)c0;
    const float *l1 = c1 < 0 ? leaf_lo + 3 * (size_t)(~c1) : node_lo + 3 * (size_t)c1, *h1 = c1 < 0 ? leaf_hi + 3 * (size_t)(~c1) : node_hi + 3 * (size_t)c1;
    DevNode d;
    d.q[0] = make_float4(l0[0], l0[1], l0[2], h0[0]);
    d.q[1] = make_float4(h0[1], h0[2], l1[0], l1[1]);
    d.q[2] = make_float4(l1[2], h1[0], h1[1], h1[2]);
    d.q[3] = make_float4(__int_as_float(c0), __int_as_float(c1), 0.f, 0.f);
    nodes[n] = d;
}

// ---- 4-wide collapse + 8-bit quantisation of the binary LBVH (host side; build time only) ----------------------
// Greedy collapse: start from a binary node's two children and keep replacing the internal candidate with the largest
// surface area by its own two children until there are four (or only leaves are left).
namespace {
struct BoxRef { const float *lo, *hi; };
inline float half_area(const float *lo, const float *hi) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}
} // namespace

// ceil(log2(x)) for x > 0, exactly (frexp is exact; log2 of a double is not guaranteed to be the same on the host and the device)
__host__ __device__ inline int ceil_log2(double x) { int ex; double m = frexp(x, &ex); return m == 0.5 ? ex - 1 : ex; }

// the collapse on the host, one thread (ArtTuning.wide_builder = 1; what round 1 shipped): the reference the device collapse below is tested against
static hipError_t wide_build_host(Lbvh &l, uint32_t T, hipStream_t s) {
    if (l.wide) return hipSuccess; // already built for this tree
    const uint32_t NI = T > 1 ? T - 1 : 0;
    std::vector<int32_t> child(NI ? (size_t)NI * 2 : 2);
    std::vector<float> nlo(NI ? (size_t)NI * 3 : 3), nhi(NI ? (size_t)NI * 3 : 3), llo((size_t)T * 3), lhi((size_t)T * 3);
    HIPQ(hipStreamSynchronize(s));
    if (NI) {
        HIPQ(hipMemcpy(child.data(), l.trav_child ? l.trav_child : l.child, (size_t)NI * 8, hipMemcpyDeviceToHost)); // collapse the tree the binary walks use
        HIPQ(hipMemcpy(nlo.data(), l.trav_child ? l.trav_lo : l.node_lo, (size_t)NI * 12, hipMemcpyDeviceToHost));
        HIPQ(hipMemcpy(nhi.data(), l.trav_child ? l.trav_hi : l.node_hi, (size_t)NI * 12, hipMemcpyDeviceToHost));
    }
    HIPQ(hipMemcpy(llo.data(), l.leaf_lo, (size_t)T * 12, hipMemcpyDeviceToHost));
    HIPQ(hipMemcpy(lhi.data(), l.leaf_hi, (size_t)T * 12, hipMemcpyDeviceToHost));
    auto box = [&](int32_t ref) -> BoxRef {
        if (ref < 0) return BoxRef{llo.data() + 3 * (size_t)(~ref), lhi.data() + 3 * (size_t)(~ref)};
        return BoxRef{nlo.data() + 3 * (size_t)ref, nhi.data() + 3 * (size_t)ref};
    };
    std::vector<DevNode4> wide;
    std::vector<DevNodeW> widef;
    std::vector<int32_t> todo; // binary node behind each wide node, in wide-index order (BFS)
    wide.reserve(NI / 2 + 2);
    if (NI == 0) todo.push_back(~0); // single triangle: a root with one leaf child
    else todo.push_back(0);
    std::vector<uint32_t> levels; size_t level_end = 0; // breadth-first: the nodes of a level are contiguous
    for (size_t w = 0; w < todo.size(); w++) {
        if (w == level_end) { levels.push_back((uint32_t)w); level_end = todo.size(); }
        int32_t cand[4]; int nc = 0;
        if (todo[w] < 0) { cand[nc++] = todo[w]; }
        else {
            cand[nc++] = child[2 * (size_t)todo[w]]; cand[nc++] = child[2 * (size_t)todo[w] + 1];
            while (nc < 4) {
                int best = -1; float ba = -1.0f;
                for (int i = 0; i < nc; i++)
                    if (cand[i] >= 0) { BoxRef b = box(cand[i]); float a = half_area(b.lo, b.hi); if (a > ba) { ba = a; best = i; } }
                if (best < 0) break;
                int32_t n = cand[best];
                cand[best] = child[2 * (size_t)n];
                cand[nc++] = child[2 * (size_t)n + 1];
            }
        }
        // the packet walk takes the children front to back along ONE axis (0,1,2,3 or 3,2,1,0 by the rays' direction sign on it): sort them by box
        // centre along the axis the centres spread most on
        uint32_t sort_axis = 0;
        {
            float best_spread = -1.0f;
            for (uint32_t k = 0; k < 3; k++) {
                float lo = INFINITY, hi = -INFINITY;
                for (int i = 0; i < nc; i++) { BoxRef b = box(cand[i]); float c = 0.5f * b.lo[k] + 0.5f * b.hi[k]; lo = std::fmin(lo, c); hi = std::fmax(hi, c); }
                if (hi - lo > best_spread) { best_spread = hi - lo; sort_axis = k; }
            }
            auto centre = [&](int32_t ref) { BoxRef b = box(ref); return 0.5f * b.lo[sort_axis] + 0.5f * b.hi[sort_axis]; };
            std::stable_sort(cand, cand + nc, [&](int32_t x, int32_t y) { return centre(x) < centre(y); });
        }
        DevNode4 d; std::memset(&d, 0, sizeof(d));
        DevNodeW dw; std::memset(&dw, 0, sizeof(dw));
        float org[3] = {INFINITY, INFINITY, INFINITY}, top[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < nc; i++) { BoxRef b = box(cand[i]); for (int k = 0; k < 3; k++) { org[k] = std::fmin(org[k], b.lo[k]); top[k] = std::fmax(top[k], b.hi[k]); } }
        d.ox = org[0]; d.oy = org[1]; d.oz = org[2];
        uint32_t ebits[3]; float scale[3];
        for (int k = 0; k < 3; k++) {
            // smallest power of two with 255 * scale >= extent (as evaluated by the device's fma), at least 2^-100
            double ext = (double)top[k] - (double)org[k];
            int e = ext > 0 ? ceil_log2(ext / 255.0) : -100;
            if (e < -100) e = -100;
            for (;;) { scale[k] = std::ldexp(1.0f, e); if (std::fmaf(255.0f, scale[k], org[k]) >= top[k]) break; e++; }
            ebits[k] = (uint32_t)(e + 127);
        }
        uint32_t mask = 0;
        for (int i = 0; i < 4; i++) {
            if (i >= nc) { d.child[i] = kAbsentChild; dw.child[i] = kAbsentChild; for (int k = 0; k < 3; k++) { dw.box[i][k] = 3.0e38f; dw.box[i][3 + k] = 3.0e38f; d.q[k] |= 255u << (8 * i); } continue; } // an absent child: its quantised box is INVERTED (lo planes 255, hi planes 0: the per-ray walk's sign-selected slab enters it after it has left it, whatever the direction); its float box a point box out at 3e38 -- every axis' entry
            // and exit distance is +-huge with the SAME sign, so neither the octant-specialised nor the general slab test lets a finite ray in (an INVERTED box passes the general one);
            // its reference is the walks' "pop" value, so a ray that passes every box (NaN: fminf / fmaxf drop a NaN operand) still cannot leave the node array
            mask |= 1u << i;
            BoxRef b = box(cand[i]);
            for (int k = 0; k < 3; k++) {
                int ql = (int)std::floor(((double)b.lo[k] - (double)org[k]) / (double)scale[k]);
                int qh = (int)std::ceil(((double)b.hi[k] - (double)org[k]) / (double)scale[k]);
                ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql); qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
                while (ql > 0 && std::fmaf((float)ql, scale[k], org[k]) > b.lo[k]) ql--;   // never taken by construction; kept as a guard
                while (qh < 255 && std::fmaf((float)qh, scale[k], org[k]) < b.hi[k]) qh++;
                d.q[k] |= (uint32_t)ql << (8 * i);
                d.q[3 + k] |= (uint32_t)qh << (8 * i);
            }
            for (int k = 0; k < 3; k++) { dw.box[i][k] = b.lo[k]; dw.box[i][3 + k] = b.hi[k]; }
            if (cand[i] < 0) d.child[i] = cand[i];
            else { d.child[i] = (int32_t)todo.size(); todo.push_back(cand[i]); }
            dw.child[i] = d.child[i];
        }
        d.exps = ebits[0] | (ebits[1] << 8) | (ebits[2] << 16) | (mask << 24);
        dw.valid = mask; dw.pad[0] = sort_axis;
        wide.push_back(d);
        widef.push_back(dw);
    }
    l.n_wide = (uint32_t)wide.size();
    levels.push_back(l.n_wide); l.wide_levels = levels;
    HIPQ(hipMalloc(&l.wide, wide.size() * sizeof(DevNode4)));
    HIPQ(hipMemcpy(l.wide, wide.data(), wide.size() * sizeof(DevNode4), hipMemcpyHostToDevice));
    HIPQ(hipMalloc(&l.widef, widef.size() * sizeof(DevNodeW)));
    HIPQ(hipMemcpy(l.widef, widef.data(), widef.size() * sizeof(DevNodeW), hipMemcpyHostToDevice));
    return hipSuccess;
}

// ---- the same collapse on the device, level by level -------------------------------------------------------------------------------------------
// A wide node is a binary node whose two children are expanded greedily (largest half-area first) to at most four.  Level L's wide nodes are the
// internal children of level L-1's, in order: the index of a child is (first index of its level) + (exclusive scan of the internal-children counts),
// which is exactly the breadth-first numbering the host loop produces -- the two builders emit the same arrays, bit for bit.
struct WideIn { const int32_t *child; const float *nlo, *nhi, *llo, *lhi; };
__device__ inline void wide_box(const WideIn &in, int32_t ref, const float *&lo, const float *&hi) {
    if (ref < 0) { lo = in.llo + 3 * (size_t)(~ref); hi = in.lhi + 3 * (size_t)(~ref); }
    else { lo = in.nlo + 3 * (size_t)ref; hi = in.nhi + 3 * (size_t)ref; }
}
// pass 1: the (sorted) children of every wide node of the level, and how many of them are internal
__global__ __launch_bounds__(256) void k_wide_expand(WideIn in, const int32_t *__restrict__ front, uint32_t n, int32_t *__restrict__ cand_out /*4 per node*/,
                                                     uint32_t *__restrict__ meta /*nc | axis << 8*/, uint32_t *__restrict__ n_internal) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n) return;
    int32_t cand[4] = {kAbsentChild, kAbsentChild, kAbsentChild, kAbsentChild}; int nc = 0;
    const int32_t root = front[w];
    if (root < 0) cand[nc++] = root; // single triangle: a root with one leaf child
    else {
        cand[nc++] = in.child[2 * (size_t)root]; cand[nc++] = in.child[2 * (size_t)root + 1];
        while (nc < 4) {
            int best = -1; float ba = -1.0f;
            for (int i = 0; i < nc; i++)
                if (cand[i] >= 0) { const float *lo, *hi; wide_box(in, cand[i], lo, hi); float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; float a = dx * dy + dy * dz + dz * dx; if (a > ba) { ba = a; best = i; } }
            if (best < 0) break;
            int32_t nn = cand[best];
            cand[best] = in.child[2 * (size_t)nn];
            cand[nc++] = in.child[2 * (size_t)nn + 1];
        }
    }
    uint32_t sort_axis = 0; float best_spread = -1.0f;
    float cen[3][4];
    for (uint32_t k = 0; k < 3; k++) {
        float lo_c = INFINITY, hi_c = -INFINITY;
        for (int i = 0; i < nc; i++) { const float *lo, *hi; wide_box(in, cand[i], lo, hi); float c = 0.5f * lo[k] + 0.5f * hi[k]; cen[k][i] = c; lo_c = fminf(lo_c, c); hi_c = fmaxf(hi_c, c); }
        if (hi_c - lo_c > best_spread) { best_spread = hi_c - lo_c; sort_axis = k; }
    }
    float key[4];
    for (int i = 0; i < nc; i++) key[i] = sort_axis == 0 ? cen[0][i] : (sort_axis == 1 ? cen[1][i] : cen[2][i]);
    for (int i = 1; i < nc; i++) // stable insertion sort by centre (std::stable_sort's order)
        for (int j = i; j > 0 && key[j] < key[j - 1]; j--) { float tk = key[j]; key[j] = key[j - 1]; key[j - 1] = tk; int32_t tc = cand[j]; cand[j] = cand[j - 1]; cand[j - 1] = tc; }
    uint32_t ni = 0;
    for (int i = 0; i < nc; i++) ni += cand[i] >= 0;
    for (int i = 0; i < 4; i++) cand_out[4 * (size_t)w + i] = cand[i];
    meta[w] = (uint32_t)nc | (sort_axis << 8);
    n_internal[w] = ni;
}
// The quantised record of a 4-wide node from its nc child boxes: origin = the boxes' common corner, one power-of-two scale per axis, 8-bit planes rounded
// outwards so that origin + q * scale -- as the walks' fma evaluates it -- contains the float box.  Shared by the collapse and by the refit.
__device__ inline void wide_quantise(const float *const lo[4], const float *const hi[4], int nc, DevNode4 &d) {
    for (int k = 0; k < 6; k++) d.q[k] = 0;
    d.spare[0] = d.spare[1] = 0;
    float org[3] = {INFINITY, INFINITY, INFINITY}, top[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < nc; i++) if (!box_nowhere(lo[i][0])) for (int k = 0; k < 3; k++) { org[k] = fminf(org[k], lo[i][k]); top[k] = fmaxf(top[k], hi[i][k]); }
    if (org[0] > top[0]) for (int k = 0; k < 3; k++) { org[k] = 0.f; top[k] = 0.f; }   // every child masked (refit): any frame of reference will do
    d.ox = org[0]; d.oy = org[1]; d.oz = org[2];
    uint32_t ebits[3]; float scale[3];
    for (int k = 0; k < 3; k++) { // smallest power of two with 255 * scale >= extent (as evaluated by the tracer's fma), at least 2^-100
        double ext = (double)top[k] - (double)org[k];
        int e = ext > 0 ? ceil_log2(ext / 255.0) : -100;
        if (e < -100) e = -100;
        for (;;) { scale[k] = ldexpf(1.0f, e); if (fmaf(255.0f, scale[k], org[k]) >= top[k]) break; e++; }
        ebits[k] = (uint32_t)(e + 127);
    }
    uint32_t mask = 0;
    for (int i = nc; i < 4; i++) for (int k = 0; k < 3; k++) d.q[k] |= 255u << (8 * i); // an absent child: an inverted box (lo planes 255, hi planes 0), which the per-ray walk's sign-selected slab never enters
    for (int i = 0; i < nc; i++) {
        mask |= 1u << i;
        if (box_nowhere(lo[i][0])) { for (int k = 0; k < 3; k++) d.q[k] |= 255u << (8 * i); continue; }   // a masked subtree: the inverted box of an absent child
        for (int k = 0; k < 3; k++) {
            int ql = (int)floor(((double)lo[i][k] - (double)org[k]) / (double)scale[k]);
            int qh = (int)ceil(((double)hi[i][k] - (double)org[k]) / (double)scale[k]);
            ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql); qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
            while (ql > 0 && fmaf((float)ql, scale[k], org[k]) > lo[i][k]) ql--;
            while (qh < 255 && fmaf((float)qh, scale[k], org[k]) < hi[i][k]) qh++;
            d.q[k] |= (uint32_t)ql << (8 * i);
            d.q[3 + k] |= (uint32_t)qh << (8 * i);
        }
    }
    d.exps = ebits[0] | (ebits[1] << 8) | (ebits[2] << 16) | (mask << 24);
}
// pass 2: the two node records of every wide node of the level; its internal children become the next level's wide nodes
__global__ __launch_bounds__(256) void k_wide_emit(WideIn in, uint32_t n, uint32_t first /*wide index of this level's first node*/, const int32_t *__restrict__ cand_in,
                                                   const uint32_t *__restrict__ meta, const uint32_t *__restrict__ offs, DevNode4 *__restrict__ wide,
                                                   DevNodeW *__restrict__ widef, int32_t *__restrict__ next_front) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n) return;
    const int nc = (int)(meta[w] & 255u); const uint32_t sort_axis = meta[w] >> 8;
    int32_t cand[4];
    for (int i = 0; i < 4; i++) cand[i] = cand_in[4 * (size_t)w + i];
    DevNode4 d; DevNodeW dw;
    dw.pad[1] = dw.pad[2] = 0;
    const float *lo[4] = {nullptr, nullptr, nullptr, nullptr}, *hi[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < nc; i++) wide_box(in, cand[i], lo[i], hi[i]);
    wide_quantise(lo, hi, nc, d);
    uint32_t next = first + n + offs[w]; // this node's internal children follow those of the nodes before it
    for (int i = 0; i < 4; i++) {
        if (i >= nc) { d.child[i] = kAbsentChild; dw.child[i] = kAbsentChild; for (int k = 0; k < 3; k++) { dw.box[i][k] = 3.0e38f; dw.box[i][3 + k] = 3.0e38f; } continue; }
        for (int k = 0; k < 3; k++) { dw.box[i][k] = lo[i][k]; dw.box[i][3 + k] = hi[i][k]; }
        if (cand[i] < 0) d.child[i] = cand[i];
        else { d.child[i] = (int32_t)next; next_front[next - (first + n)] = cand[i]; next++; }
        dw.child[i] = d.child[i];
    }
    dw.valid = d.exps >> 24; dw.pad[0] = sort_axis;
    wide[first + w] = d;
    widef[first + w] = dw;
}

hipError_t wide_build(Lbvh &l, uint32_t T, hipStream_t s, bool on_host) {
    if (l.wide) return hipSuccess; // already built for this tree
    if (on_host) return wide_build_host(l, T, s);
    const uint32_t NI = T > 1 ? T - 1 : 0;
    const uint32_t cap = NI ? NI : 1;   // a wide node stands on a distinct binary node: at most NI of them (one for a single triangle)
    WideIn in{l.trav_child ? l.trav_child : l.child, l.trav_child ? l.trav_lo : l.node_lo, l.trav_child ? l.trav_hi : l.node_hi, l.leaf_lo, l.leaf_hi};
    int32_t *front[2] = {nullptr, nullptr}, *cand = nullptr; uint32_t *meta = nullptr, *cnt = nullptr, *offs = nullptr; void *tmp = nullptr; size_t tmp_bytes = 0;
    DevNode4 *wide = nullptr; DevNodeW *widef = nullptr;
    std::vector<uint32_t> levels; // first node of every level (breadth-first numbering: a level is contiguous), then n_wide
    Arena own; Arena &A = l.arena ? *l.arena : own;
    auto body = [&]() -> hipError_t {
        HIPQ(rocprim::exclusive_scan(nullptr, tmp_bytes, cnt, offs, 0u, (size_t)cap + 1, rocprim::plus<uint32_t>(), s));
        HIPQ(A.reserve(3 * Arena::pad((size_t)cap * 4) + Arena::pad((size_t)cap * 16) + 2 * Arena::pad(((size_t)cap + 1) * 4) + Arena::pad((size_t)cap * sizeof(DevNode4)) + Arena::pad((size_t)cap * sizeof(DevNodeW)) + Arena::pad(tmp_bytes ? tmp_bytes : 16)));
        front[0] = A.take<int32_t>(cap); front[1] = A.take<int32_t>(cap); cand = A.take<int32_t>((size_t)cap * 4);
        meta = A.take<uint32_t>(cap); cnt = A.take<uint32_t>((size_t)cap + 1); offs = A.take<uint32_t>((size_t)cap + 1);
        wide = A.take<DevNode4>(cap); widef = A.take<DevNodeW>(cap);
        tmp = A.take<char>(tmp_bytes ? tmp_bytes : 16);
        const int32_t root = NI ? 0 : ~0;
        HIPQ(hipMemcpyAsync(front[0], &root, 4, hipMemcpyHostToDevice, s));
        uint32_t first = 0, n = 1; int cur = 0;
        while (n) {
            if (first + n > cap) return hipErrorInvalidValue; // (cannot happen: see cap)
            const uint32_t g = (n + 255) / 256;
            k_wide_expand<<<g, 256, 0, s>>>(in, front[cur], n, cand, meta, cnt);
            HIPQ(hipMemsetAsync(cnt + n, 0, 4, s));                                  // the scan's last element = the level's total
            size_t tb = tmp_bytes;
            HIPQ(rocprim::exclusive_scan(tmp, tb, cnt, offs, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), s));
            k_wide_emit<<<g, 256, 0, s>>>(in, n, first, cand, meta, offs, wide, widef, front[cur ^ 1]);
            uint32_t n_next = 0;
            HIPQ(hipMemcpyAsync(&n_next, offs + n, 4, hipMemcpyDeviceToHost, s));
            HIPQ(hipStreamSynchronize(s));
            levels.push_back(first);
            first += n; n = n_next; cur ^= 1;
        }
        HIPQ(hipGetLastError());
        l.n_wide = first;
        levels.push_back(first); l.wide_levels = levels;
        return hipSuccess;
    };
    hipError_t e = body();
    if (e == hipSuccess) { // the work arrays are sized for the worst case (NI nodes); a 4-wide collapse of a binary tree has about a third of that
        e = hipMalloc(&l.wide, (size_t)l.n_wide * sizeof(DevNode4));
        if (e == hipSuccess) e = hipMalloc(&l.widef, (size_t)l.n_wide * sizeof(DevNodeW));
        if (e == hipSuccess) e = hipMemcpyAsync(l.wide, wide, (size_t)l.n_wide * sizeof(DevNode4), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(l.widef, widef, (size_t)l.n_wide * sizeof(DevNodeW), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { hipFree(l.wide); hipFree(l.widef); l.wide = nullptr; l.widef = nullptr; l.n_wide = 0; }
    }
    own.release();
    return e;
}

// ---- refit: a model moved (VkModel::set_model_matrix, vk_model.rs:461-466; the reference rebuilds its TLAS every frame for this, renderer.rs:637-651) ------
// The topology stays, the world-space triangles of the version and every box above them are made again.  Results cannot depend on it: accept() is per
// triangle and every node box is an exact min / max union of the triangle boxes below it -- the frames are those of a fresh build.

// the version's triangle records from the object-space shading records (the vertices k_leaves gathered) and the version's object->world matrices: the very
// operations of k_soup (transform) and k_leaves (edges, box), so a refit with unchanged matrices writes the bits that are there
// Only what moved is made again: touched[primitive] says whose triangles (the primitives moved since this VERSION was last written).  The primitive table and
// the touched bytes are read WHERE THE HOST WROTE THEM (pinned, device-visible memory: a few hundred bytes a wave touches once) -- round 3 uploaded both with two
// copies in front of this launch; the launch also leaves the table's device copy for the frames behind it (their shading reads DevPrim).
// A rewritten triangle marks the 4-wide node that holds it (mark[w] != 0: a box below w changes in this refit); the level passes below carry the marks upwards.
__device__ __forceinline__ void retri_one(uint32_t p, const DevShadeTri *__restrict__ shade, const DevPrim *prims_host, const uint8_t *touched, const uint32_t *__restrict__ leaf_parent, uint32_t *mark, DevTri *tris) {
    const uint32_t prim = __float_as_uint(shade[p].f[34]);
    if (!touched[prim]) return;
    DevTri t;
    if (prims_host[prim].masked) {   // out of the structure until it is enabled again: a point nowhere, no extent
        for (int k = 0; k < 3; k++) { t.f[k] = kNowhere; t.f[3 + k] = 0.f; t.f[6 + k] = 0.f; t.f[9 + k] = kNowhere; t.f[12 + k] = kNowhere; }
    } else {
        const float4 *sq = reinterpret_cast<const float4 *>(shade + p);
        const float4 s0 = sq[0], s1 = sq[1], s2 = sq[2];
        const float *m = prims_host[prim].o2w;
        const float3 w0 = xform_point(m, s0.x, s0.y, s0.z), w1 = xform_point(m, s0.w, s1.x, s1.y), w2 = xform_point(m, s1.z, s1.w, s2.x);
        const float a[3] = {w0.x, w0.y, w0.z}, b[3] = {w1.x, w1.y, w1.z}, c[3] = {w2.x, w2.y, w2.z};
        for (int k = 0; k < 3; k++) {
            t.f[k] = a[k]; t.f[3 + k] = b[k] - a[k]; t.f[6 + k] = c[k] - a[k];
            t.f[9 + k] = fminf(fminf(a[k], b[k]), c[k]); t.f[12 + k] = fmaxf(fmaxf(a[k], b[k]), c[k]);
        }
    }
    t.f[15] = tris[p].f[15]; // the global triangle id never changes
    tris[p] = t;
    mark[leaf_parent[p]] = 1u;
}
// which 4-wide node holds a leaf, which one a node (the root: ~0): made once per tree, when its first model moves
__global__ __launch_bounds__(256) void k_wide_parents(uint32_t n_wide, const DevNodeW *__restrict__ widef, uint32_t *__restrict__ leaf_parent, uint32_t *__restrict__ node_parent) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_wide) return;
    if (w == 0) node_parent[0] = ~0u;
    for (int i = 0; i < 4; i++) {
        const int32_t ch = widef[w].child[i];
        if (ch == kAbsentChild) continue;
        if (ch < 0) leaf_parent[(uint32_t)~ch] = w; else node_parent[ch] = w;
    }
}

// One child box of one node of the 4-wide tree: a leaf child's box is its triangle's, an internal child's the union of that node's own child boxes (float min / max
// are exact, so whatever the order the union is the box a build would find).  Four neighbouring lanes share a node; the pass is loads and min / max only -- the
// quantised records are made afterwards, for all nodes at once (k_wide_requant): their double-precision arithmetic does not belong on the bottom-up critical path.
__device__ __forceinline__ void wide_union_child(uint32_t w, uint32_t i, const DevTri *__restrict__ tris, DevNodeW *widef, const uint32_t *__restrict__ node_parent, uint32_t *mark) {
    if (!mark[w]) return;                               // nothing below this node moved
    if (i == 0) { const uint32_t up = node_parent[w]; if (up != ~0u) mark[up] = 1u; }   // its parent's box of it changes: a level further up, a launch (or a barrier) later
    const int32_t ch = widef[w].child[i];
    if (ch == kAbsentChild) return;
    float lo[3], hi[3];
    if (ch < 0) {
        const float4 *tq = reinterpret_cast<const float4 *>(tris + (uint32_t)~ch);
        const float4 c = tq[2], d = tq[3];                  // f[8..11] | f[12..15]: e2.z lo.xyz | hi.xyz gid
        lo[0] = c.y; lo[1] = c.z; lo[2] = c.w; hi[0] = d.x; hi[1] = d.y; hi[2] = d.z;
    } else {
        const float4 *q = reinterpret_cast<const float4 *>(widef + ch);
        const float4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4], a5 = q[5];
        const uint32_t cv = widef[ch].valid;
        const float b[4][6] = {{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y}, {a1.z, a1.w, a2.x, a2.y, a2.z, a2.w}, {a3.x, a3.y, a3.z, a3.w, a4.x, a4.y}, {a4.z, a4.w, a5.x, a5.y, a5.z, a5.w}};
        for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
        for (int j = 0; j < 4; j++) if (((cv >> j) & 1u) && !box_nowhere(b[j][0])) for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], b[j][k]); hi[k] = fmaxf(hi[k], b[j][3 + k]); }
        if (lo[0] > hi[0]) for (int k = 0; k < 3; k++) { lo[k] = kNowhere; hi[k] = kNowhere; }   // nothing left below that node: nowhere itself
    }
    float *o = widef[w].box[i];
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
}
// one node of the pass behind the boxes: its quantised record (the per-ray walks' DevNode4) from its float one if asked, and what it adds to the tree's surface-area cost
// (the half-areas of its child boxes); the root also leaves the half-area of its union in acc[1]
__device__ __forceinline__ double wide_requant_node(uint32_t w, const DevNodeW *__restrict__ widef, DevNode4 *__restrict__ wide, bool requant, double *acc) {
    const float4 *q = reinterpret_cast<const float4 *>(widef + w);
    const float4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4], a5 = q[5];
    const int4 ch = *reinterpret_cast<const int4 *>(&widef[w].child[0]);
    const float lo[4][3] = {{a0.x, a0.y, a0.z}, {a1.z, a1.w, a2.x}, {a3.x, a3.y, a3.z}, {a4.z, a4.w, a5.x}}, hi[4][3] = {{a0.w, a1.x, a1.y}, {a2.y, a2.z, a2.w}, {a3.w, a4.x, a4.y}, {a5.y, a5.z, a5.w}};
    const int nc = (ch.x != kAbsentChild) + (ch.y != kAbsentChild) + (ch.z != kAbsentChild) + (ch.w != kAbsentChild); // the valid children are slots 0 .. nc-1
    if (requant) {
        const float *plo[4] = {lo[0], lo[1], lo[2], lo[3]}, *phi[4] = {hi[0], hi[1], hi[2], hi[3]};
        DevNode4 d;
        wide_quantise(plo, phi, nc, d);
        d.child[0] = ch.x; d.child[1] = ch.y; d.child[2] = ch.z; d.child[3] = ch.w;
        wide[w] = d;
    }
    double a = 0.0;
    float rlo[3] = {INFINITY, INFINITY, INFINITY}, rhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int j = 0; j < nc; j++) {
        if (box_nowhere(lo[j][0])) continue;
        double dx = (double)hi[j][0] - lo[j][0], dy = (double)hi[j][1] - lo[j][1], dz = (double)hi[j][2] - lo[j][2];
        a += dx * dy + dy * dz + dz * dx;
        for (int k = 0; k < 3; k++) { rlo[k] = fminf(rlo[k], lo[j][k]); rhi[k] = fmaxf(rhi[k], hi[j][k]); }
    }
    if (w == 0) { double dx = (double)rhi[0] - rlo[0], dy = (double)rhi[1] - rlo[1], dz = (double)rhi[2] - rlo[2]; acc[1] = dx < 0 ? 0.0 : dx * dy + dy * dz + dz * dx; }
    return a;
}
// The tree cut into BATCHES of whole subtrees of about kBatchNodes nodes (refit_lists_build) and the crown above them: ONE workgroup rewrites a batch's triangles and then refits
// its nodes level by level, deepest first, with a workgroup barrier between levels -- everything a node of the batch depends on is the batch's own, written through the one L1
// of the CU the workgroup runs on; a second launch of one large workgroup does the same for the crown (the few hundred nodes whose subtrees are larger than a batch).
// Round 3 ran a launch per level of the whole tree (a dozen dependent launches in front of every frame of a moving model: among three frames' worth of waves each waits tens
// of microseconds for its turn); round 4 first tried the marked nodes bottom-up in ONE launch with arrival counters -- release / atomic / acquire per node, k_refit's
// protocol -- and measured 0.41 ms for that launch alone and 1.0 ms among frames: an agent-scope release on gfx950 writes the XCD's L2 back and every acquire invalidates
// it, per node and level, where a workgroup barrier costs nothing of the kind.  profiles/README.md round 4.
// sub_off: [0, nb + 1] node offsets of batches 0 .. nb (batch nb = the crown) | [nb + 2, 2 nb + 3] leaf offsets | then nb + 1 rows of (n_levels + 1) offsets into the batch's
// node list, deepest level of the tree first.
template <bool FOLD> __global__ void k_refit_sub(uint32_t batch0, uint32_t nb1 /*batches + the crown*/, uint32_t n_levels, const uint32_t *__restrict__ sub_nodes, const uint32_t *__restrict__ sub_leaves, const uint32_t *__restrict__ sub_off,
                            const DevShadeTri *__restrict__ shade, const DevPrim *prims_host, DevPrim *prims_dev, uint32_t n_prim_words, const uint8_t *touched,
                            const uint32_t *__restrict__ leaf_parent, const uint32_t *__restrict__ node_parent, uint32_t *mark, DevTri *tris, DevNodeW *widef, unsigned long long *stamp, bool first_launch,
                            DevNode4 *wide, double *acc, double *out /*null: not the refit's last launch*/, const uint32_t *__restrict__ dirty /*null: batch batch0 + blockIdx.x*/, double *batch_cost) {
    const uint32_t b = dirty ? dirty[blockIdx.x] : batch0 + blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    if (first_launch) {
        if (blockIdx.x == 0 && tid == 0) stamp[0] = wall_clock64();   // the refit's start (100 MHz): its device time travels to the host with its cost, no events
        for (uint32_t i = blockIdx.x * nt + tid; i < n_prim_words; i += gridDim.x * nt) reinterpret_cast<uint32_t *>(prims_dev)[i] = reinterpret_cast<const uint32_t *>(prims_host)[i];   // the table's device copy, for the frames
    }
    for (uint32_t i = sub_off[nb1 + 1 + b] + tid; i < sub_off[nb1 + 2 + b]; i += nt) retri_one(sub_leaves[i], shade, prims_host, touched, leaf_parent, mark, tris);
    __threadfence_block();
    __syncthreads();
    const uint32_t *lv_off = sub_off + 2 * (nb1 + 1) + (size_t)b * (n_levels + 1), n0 = sub_off[b];
    for (uint32_t lv = 0; lv < n_levels; lv++) {         // (the same trips for every thread of the workgroup)
        const uint32_t lo = lv_off[lv], hi = lv_off[lv + 1];
        if (hi == lo) continue;
        for (uint32_t i = 4u * lo + tid; i < 4u * hi; i += nt) wide_union_child(sub_nodes[n0 + (i >> 2)], i & 3u, tris, widef, node_parent, mark);
        __threadfence_block();
        __syncthreads();   // the level above reads these records
    }
    // Round 4g: what k_wide_requant did in a launch of its own behind the crown -- 514 workgroups finding their slots among the frames' packets, 0.2-0.3 ms of the refit's
    // 0.5-0.7 -- each workgroup now does for ITS nodes while they are warm: the quantised records of the marked ones (marks cleared), its share of the tree's cost.  The
    // crown's workgroup runs behind all the batches (stream order) and hands the result to the host.
    if (!FOLD) return;   // (a small tree: k_wide_requant does this behind the crown, a node a thread -- see launch_refit; the instance without this tail keeps 40 registers
                         //  and eight waves a SIMD -- with it 122, and a crown of 1 024 threads would need a CU all to itself)
    __shared__ double s_cost[16];
    double a = 0.0;
    for (uint32_t i = sub_off[b] + tid; i < sub_off[b + 1]; i += nt) {
        const uint32_t w = sub_nodes[i];
        const bool m = mark[w] != 0u;
        if (m) mark[w] = 0u;
        a += wide_requant_node(w, widef, wide, m && wide, acc);
    }
    if (out) for (uint32_t i = tid; i + 1u < nb1; i += nt) a += batch_cost[i];   // the crown: every batch's share, made by this refit or kept from an earlier one of this version
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off);
    if ((tid & 63u) == 0) s_cost[tid >> 6] = a;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (uint32_t i = 0; i < (nt + 63u) / 64u; i++) t += s_cost[i];
        if (out) {
            out[0] = t; out[1] = *(volatile double *)&acc[1];
            reinterpret_cast<unsigned long long *>(out)[2] = *(volatile unsigned long long *)&acc[2]; reinterpret_cast<unsigned long long *>(out)[3] = wall_clock64();
            acc[0] = 0.0; acc[1] = 0.0;
        } else batch_cost[b] = t;
    }
}
// after the boxes: every node's quantised record (the per-ray walks', DevNode4) from its float one, and the tree's surface-area cost while the boxes are at hand:
// cost[0] += the half-areas of all child boxes (the measure of the rays that cross each box: what a walk pays for), cost[1] = half-area of the root's union.  A refit
// can only keep or grow the sum against the same rays; a rebuild restores it.  The rule compares the plain sums: dividing by the root's area would reward a model
// that flies off (the root grows faster than the sum) although the rays of a camera among the rest of the scene cross more boxes than before.
// acc (device): [0] the running sum, [1] the root's half-area, [2] the refit's start stamp (k_retri), [3] exit ticket of this launch (as a 64-bit counter).  The last block
// out writes {sum, root, start, end} to `out` -- pinned host memory when a refit's result travels to the host behind its `ready` event (no copy, no event of its
// own), a device buffer otherwise -- and clears acc for the next launch.
__global__ __launch_bounds__(256) void k_wide_requant(uint32_t n_wide, const DevNodeW *__restrict__ widef, DevNode4 *__restrict__ wide, uint32_t *mark /*null: every node*/, double *acc, double *out) {
    __shared__ double s_part[4];
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    double a = 0.0;
    if (w < n_wide) {
        const bool requant = wide && (!mark || mark[w]);      // the double-precision quantisation only where a box changed; the cost sums every node
        if (mark && mark[w]) mark[w] = 0;                      // (the marks are this version's: cleared for its next refit)
        a = wide_requant_node(w, widef, wide, requant, acc);
    }
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off);
    if ((threadIdx.x & 63u) == 0) s_part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        if (t != 0.0) atomicAdd(&acc[0], t);
        __threadfence();
        unsigned long long *ticket = reinterpret_cast<unsigned long long *>(&acc[3]);
        if (atomicAdd(ticket, 1ull) == (unsigned long long)gridDim.x - 1ull) {   // every block's sum (and block 0's root area) is in
            __threadfence();
            out[0] = atomicAdd(&acc[0], 0.0); out[1] = *(volatile double *)&acc[1];
            reinterpret_cast<unsigned long long *>(out)[2] = *(volatile unsigned long long *)&acc[2]; reinterpret_cast<unsigned long long *>(out)[3] = wall_clock64();
            acc[0] = 0.0; acc[1] = 0.0; *ticket = 0ull;
        }
    }
}

void launch_wide_parents(uint32_t n_wide, const DevNodeW *widef, uint32_t *leaf_parent, uint32_t *node_parent, hipStream_t s) {
    k_wide_parents<<<(n_wide + 255) / 256, 256, 0, s>>>(n_wide, widef, leaf_parent, node_parent);
}
// a refit in two or three launches whatever the tree's depth: the batches (their triangles, their nodes' boxes), the crown (the same for the nodes above the batches), and the
// quantised records + cost either inside those (a large tree) or in a launch behind them -> result[0..3] = cost sum, root half-area, start and end stamps (wall_clock64: 100 MHz)
void launch_refit(const RefitArgs &r, hipStream_t s) {
    unsigned long long *stamp = reinterpret_cast<unsigned long long *>(r.acc) + 2;
    const uint32_t npw = r.n_prims * (uint32_t)(sizeof(DevPrim) / 4), nb1 = r.sub_batches + 1;
    // The quantised records and the cost: in the refit's own workgroups for a large tree (config 4's 1.4 M nodes: a refit alone 0.57 -> 0.28 ms, a frame of a moving model
    // 0.67 -> 0.44), in a launch of their own -- a node a thread -- for a small one (config 2's 131 k nodes: folded, a refit alone 0.13 -> 0.19 ms: three nodes' double-precision
    // quantisation in a row per thread of 143 workgroups against one each in 514; among frames the two forms cost the same).  profiles/README.md round 4g
    const uint32_t grid = r.dirty ? r.n_dirty : r.sub_batches;   // (a refit runs the batches that hold a primitive that moved; the crown always)
    if (r.fold) {
        if (grid) k_refit_sub<true><<<grid, 256, 0, s>>>(0u, nb1, r.sub_levels, r.sub_nodes, r.sub_leaves, r.sub_off, r.shade, r.prims_host, r.prims_dev, npw, r.touched, r.leaf_parent, r.node_parent, r.mark, r.tris, r.widef, stamp, true, r.wide, r.acc, nullptr, r.dirty, r.batch_cost);
        k_refit_sub<true><<<1, r.sub_batches ? 256 : 1024, 0, s>>>(r.sub_batches, nb1, r.sub_levels, r.sub_nodes, r.sub_leaves, r.sub_off, r.shade, r.prims_host, r.prims_dev, npw, r.touched, r.leaf_parent, r.node_parent, r.mark, r.tris, r.widef, stamp, grid == 0, r.wide, r.acc, r.result, nullptr, r.batch_cost);
        return;
    }
    if (grid) k_refit_sub<false><<<grid, 256, 0, s>>>(0u, nb1, r.sub_levels, r.sub_nodes, r.sub_leaves, r.sub_off, r.shade, r.prims_host, r.prims_dev, npw, r.touched, r.leaf_parent, r.node_parent, r.mark, r.tris, r.widef, stamp, true, nullptr, nullptr, nullptr, r.dirty, nullptr);
    k_refit_sub<false><<<1, 1024, 0, s>>>(r.sub_batches, nb1, r.sub_levels, r.sub_nodes, r.sub_leaves, r.sub_off, r.shade, r.prims_host, r.prims_dev, npw, r.touched, r.leaf_parent, r.node_parent, r.mark, r.tris, r.widef, stamp, grid == 0, nullptr, nullptr, nullptr, nullptr, nullptr);
    k_wide_requant<<<(r.n_wide + 255) / 256, 256, 0, s>>>(r.n_wide, r.widef, r.wide, r.mark, r.acc, r.result);
}
// The refit's work lists, once per tree (host work on the parents read back: the topology never changes).  A node whose subtree has at most kBatchNodes nodes while its parent's
// has more roots a batch subtree; runs of such roots (in index order) are dealt to batches of about kBatchNodes nodes; the nodes above them -- the crown: a few hundred for the
// bench scenes -- are batch `nb`.  The numbering is breadth-first, so a larger index is never above a smaller one: a batch's nodes in descending order are deepest level first.
__global__ __launch_bounds__(256) void k_leaf_prims(uint32_t T, const uint32_t *__restrict__ leaf_gid, const uint32_t *__restrict__ tri_prim, uint32_t *__restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < T) out[p] = tri_prim[leaf_gid[p]];
}
hipError_t refit_lists_build(Lbvh &l, uint32_t T, hipStream_t s) {
    constexpr uint32_t kBatchNodes = 768;
    const std::vector<uint32_t> &levels = l.wide_levels;
    const uint32_t NW = l.n_wide, n_levels = (uint32_t)levels.size() - 1;
    std::vector<uint32_t> node_parent(NW), leaf_parent(T);
    HIPQ(hipStreamSynchronize(s));
    HIPQ(hipMemcpy(node_parent.data(), l.node_parent, (size_t)NW * 4, hipMemcpyDeviceToHost)); HIPQ(hipMemcpy(leaf_parent.data(), l.leaf_parent, (size_t)T * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> size(NW, 1);
    for (uint32_t w = NW; w-- > 1;) size[node_parent[w]] += size[w];
    // batch_of: ~0 = the crown; roots in index order fill batches
    std::vector<uint32_t> batch_of(NW, ~0u);
    uint32_t nb = 0, fill = 0;
    for (uint32_t w = 1; w < NW; w++) {
        if (size[w] > kBatchNodes) continue;                                   // crown
        const uint32_t up = node_parent[w];
        if (size[up] > kBatchNodes) {                                          // a batch root
            if (nb == 0 || fill + size[w] > kBatchNodes + kBatchNodes / 2) { nb++; fill = 0; }
            fill += size[w]; batch_of[w] = nb - 1;
        } else batch_of[w] = batch_of[up];
    }
    const uint32_t nb1 = nb + 1;
    auto slot = [&](uint32_t w) { return batch_of[w] == ~0u ? nb : batch_of[w]; };
    std::vector<uint32_t> off(2 * (size_t)(nb1 + 1) + (size_t)nb1 * (n_levels + 1), 0), sub_nodes(NW), sub_leaves(T);
    {   // node lists: counts -> offsets -> fill from the deepest level up; level offsets within each list
        std::vector<uint32_t> cnt(nb1 + 1, 0);
        for (uint32_t w = 0; w < NW; w++) cnt[slot(w) + 1]++;
        for (uint32_t b = 0; b < nb1; b++) cnt[b + 1] += cnt[b];
        for (uint32_t b = 0; b <= nb1; b++) off[b] = cnt[b];
        std::vector<uint32_t> at(cnt.begin(), cnt.end() - 1);
        uint32_t *lv_off = off.data() + 2 * (size_t)(nb1 + 1);
        for (uint32_t lv = n_levels; lv-- > 0;) {          // deepest level first
            for (uint32_t b = 0; b < nb1; b++) lv_off[(size_t)b * (n_levels + 1) + (n_levels - 1 - lv)] = at[b] - cnt[b];
            for (uint32_t w = levels[lv + 1]; w-- > levels[lv];) sub_nodes[at[slot(w)]++] = w;
        }
        for (uint32_t b = 0; b < nb1; b++) lv_off[(size_t)b * (n_levels + 1) + n_levels] = at[b] - cnt[b];
    }
    {   // leaf lists
        std::vector<uint32_t> cnt(nb1 + 1, 0);
        for (uint32_t p = 0; p < T; p++) cnt[slot(leaf_parent[p]) + 1]++;
        for (uint32_t b = 0; b < nb1; b++) cnt[b + 1] += cnt[b];
        for (uint32_t b = 0; b <= nb1; b++) off[nb1 + 1 + b] = cnt[b];
        std::vector<uint32_t> at(cnt.begin(), cnt.end() - 1);
        for (uint32_t p = 0; p < T; p++) sub_leaves[at[slot(leaf_parent[p])]++] = p;
    }
    {   // which primitives have triangles in which batch (the crown always runs): the leaf's primitive is tri_prim[leaf_gid[p]], gathered on the device
        std::vector<uint32_t> tp(T);
        uint32_t *d_lp = nullptr;
        HIPQ(hipMalloc(&d_lp, (size_t)T * 4));
        k_leaf_prims<<<(T + 255) / 256, 256, 0, s>>>(T, l.leaf_gid, l.tri_prim, d_lp);
        hipError_t ec = hipMemcpyAsync(tp.data(), d_lp, (size_t)T * 4, hipMemcpyDeviceToHost, s);   // (on the kernel's own stream: the context's streams do not wait for the null stream or it for them)
        if (ec == hipSuccess) ec = hipStreamSynchronize(s);
        hipFree(d_lp);
        HIPQ(ec);
        uint32_t n_prims = 0;
        for (uint32_t g = 0; g < T; g++) n_prims = std::max(n_prims, tp[g] + 1u);
        std::vector<uint32_t> seen(n_prims, 0u);   // seen[primitive] = batch + 1 of the last batch it was listed for (the leaf lists are batch by batch: one pass)
        l.batch_prim_off.assign(1, 0u); l.batch_prim_ids.clear();
        for (uint32_t b = 0; b < nb; b++) {
            for (uint32_t i = off[nb1 + 1 + b]; i < off[nb1 + 2 + b]; i++) {
                const uint32_t pr = tp[sub_leaves[i]];
                if (seen[pr] != b + 1u) { seen[pr] = b + 1u; l.batch_prim_ids.push_back(pr); }
            }
            l.batch_prim_off.push_back((uint32_t)l.batch_prim_ids.size());
        }
    }
    hipFree(l.sub_nodes); hipFree(l.sub_leaves); hipFree(l.sub_off); l.sub_nodes = l.sub_leaves = l.sub_off = nullptr;
    HIPQ(hipMalloc(&l.sub_nodes, (size_t)NW * 4)); HIPQ(hipMalloc(&l.sub_leaves, (size_t)T * 4)); HIPQ(hipMalloc(&l.sub_off, off.size() * 4));
    HIPQ(hipMemcpy(l.sub_nodes, sub_nodes.data(), (size_t)NW * 4, hipMemcpyHostToDevice)); HIPQ(hipMemcpy(l.sub_leaves, sub_leaves.data(), (size_t)T * 4, hipMemcpyHostToDevice));
    HIPQ(hipMemcpy(l.sub_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    l.sub_batches = nb; l.sub_levels = n_levels;
    if (l.log & 1u) {
        uint32_t crown = 0, big = 0; for (uint32_t w = 0; w < NW; w++) crown += batch_of[w] == ~0u;
        for (uint32_t b = 0; b < nb; b++) big = std::max(big, off[b + 1] - off[b]);
        std::fprintf(stderr, "[art] refit lists: %u nodes in %u levels; %u batches (largest %u nodes), a crown of %u nodes\n", NW, n_levels, nb, big, crown);
    }
    return hipSuccess;
}
// the quantised records (wide; null: leave them) and the cost of the float records as they are: result[0..1] (acc: 4 zeroed doubles of scratch)
void launch_wide_cost(uint32_t n_wide, const DevNodeW *widef, DevNode4 *wide, double *acc, double *result, hipStream_t s) {
    k_wide_requant<<<(n_wide + 255) / 256, 256, 0, s>>>(n_wide, widef, wide, nullptr, acc, result);
}

// The binary trees (the canonical LBVH of art_get_lbvh, the traversal tree of the per-ray / binary walks) after a refit: leaf boxes from the triangle records,
// node boxes bottom-up with the build's own kernel, the 64-byte node records again.  Only the non-default forms and the parity surface read them, so this runs
// on demand and under full synchronisation (art_api.hip).
__global__ __launch_bounds__(256) void k_leaf_boxes(uint32_t T, const DevTri *__restrict__ tris, float *__restrict__ leaf_lo, float *__restrict__ leaf_hi) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= T) return;
    for (int k = 0; k < 3; k++) { leaf_lo[3 * (size_t)p + k] = tris[p].f[9 + k]; leaf_hi[3 * (size_t)p + k] = tris[p].f[12 + k]; }
}
__global__ __launch_bounds__(256) void k_parents(uint32_t NI, const int32_t *__restrict__ child, int32_t *__restrict__ parent_int, int32_t *__restrict__ parent_leaf) {
    uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= NI) return;
    if (n == 0) parent_int[0] = -1;
    for (int c = 0; c < 2; c++) { int32_t ch = child[2 * (size_t)n + c]; if (ch < 0) parent_leaf[~ch] = (int32_t)n; else parent_int[ch] = (int32_t)n; }
}
hipError_t binary_refit(Lbvh &l, uint32_t T, const DevTri *tris, hipStream_t s) {
    const uint32_t NI = T > 1 ? T - 1 : 0, B = 256, GT = (T + B - 1) / B;
    k_leaf_boxes<<<GT, B, 0, s>>>(T, tris, l.leaf_lo, l.leaf_hi);
    int32_t *parent_int = nullptr, *parent_leaf = nullptr; uint32_t *arrive = nullptr;
    auto body = [&]() -> hipError_t {
        if (NI) {
            HIPQ(hipMalloc(&parent_int, (size_t)NI * 4)); HIPQ(hipMalloc(&parent_leaf, (size_t)T * 4)); HIPQ(hipMalloc(&arrive, (size_t)NI * 4));
            for (int tree = 0; tree < 2; tree++) {
                const int32_t *child = tree == 0 ? l.child : l.trav_child;
                float *nlo = tree == 0 ? l.node_lo : l.trav_lo, *nhi = tree == 0 ? l.node_hi : l.trav_hi;
                if (!child) continue;
                HIPQ(hipMemsetAsync(arrive, 0, (size_t)NI * 4, s));
                k_parents<<<(NI + B - 1) / B, B, 0, s>>>(NI, child, parent_int, parent_leaf);
                k_refit<<<GT, B, 0, s>>>(T, child, parent_int, parent_leaf, l.leaf_lo, l.leaf_hi, nlo, nhi, arrive);
            }
        }
        const uint32_t NN = NI ? NI : 1;
        if (l.trav_child) k_emit_nodes<<<(NN + B - 1) / B, B, 0, s>>>(T, l.trav_child, l.trav_lo, l.trav_hi, l.leaf_lo, l.leaf_hi, l.nodes);
        else k_emit_nodes<<<(NN + B - 1) / B, B, 0, s>>>(T, l.child, l.node_lo, l.node_hi, l.leaf_lo, l.leaf_hi, l.nodes);
        HIPQ(hipGetLastError());
        HIPQ(hipStreamSynchronize(s));
        l.canon_boxes = true;
        return hipSuccess;
    };
    hipError_t e = body();
    hipFree(parent_int); hipFree(parent_leaf); hipFree(arrive);
    return e;
}

__global__ void k_build_noop() {}
void build_prewarm(hipStream_t s) { k_build_noop<<<1, 1, 0, s>>>(); }

void lbvh_free(Lbvh &l) {
    hipFree(l.wide); hipFree(l.widef); hipFree(l.leaf_parent); hipFree(l.node_parent);
    hipFree(l.sub_nodes); hipFree(l.sub_leaves); hipFree(l.sub_off);
    if (l.trav_child != l.res_trav_child) { hipFree(l.trav_child); hipFree(l.trav_lo); hipFree(l.trav_hi); }   // (allocations of their own: no room was reserved)
    hipFree(l.block);   // leaf_gid .. shade_tris, cbounds, res_trav_*
    l = Lbvh{};
}
hipError_t lbvh_claim_trav(Lbvh &l, uint32_t NI) {
    if (l.trav_child) return hipSuccess;
    if (l.res_trav_child) { l.trav_child = l.res_trav_child; l.trav_lo = l.res_trav_lo; l.trav_hi = l.res_trav_hi; return hipSuccess; }
    hipError_t e = hipMalloc(&l.trav_child, (size_t)NI * 8);
    if (e == hipSuccess) e = hipMalloc(&l.trav_lo, (size_t)NI * 12);
    if (e == hipSuccess) e = hipMalloc(&l.trav_hi, (size_t)NI * 12);
    return e;
}

hipError_t lbvh_build(const BuildInputs &in, Lbvh &out, hipStream_t s, bool node_boxes) {
    const uint32_t T = in.T;
    const uint32_t NI = T > 1 ? T - 1 : 1;
    Arena *const ctx_arena = out.arena;
    out = Lbvh{};
    out.arena = ctx_arena;
    Arena own; Arena &A = ctx_arena ? *ctx_arena : own;
    float *triw = nullptr, *tlo = nullptr, *thi = nullptr;
    uint32_t *cslots = nullptr, *gid_in = nullptr, *arrive = nullptr;
    uint64_t *keys_in = nullptr;
    int32_t *parent_int = nullptr, *parent_leaf = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    hipError_t err = hipSuccess;
    auto body = [&]() -> hipError_t {
        // the temporaries: one reservation of the context's arena (the sort's own scratch included)
        HIPQ(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, out.keys, gid_in, out.leaf_gid, T, 0, 64, s));
        HIPQ(A.reserve(Arena::pad((size_t)T * 36) + 2 * Arena::pad((size_t)T * 12) + Arena::pad(kCbSlots * 32 * 4) + Arena::pad((size_t)T * 4) + Arena::pad((size_t)T * 8) + 2 * Arena::pad((size_t)NI * 4) + Arena::pad((size_t)T * 4) + Arena::pad(tmp_bytes ? tmp_bytes : 16)));
        triw = A.take<float>((size_t)T * 9); tlo = A.take<float>((size_t)T * 3); thi = A.take<float>((size_t)T * 3);
        cslots = A.take<uint32_t>(kCbSlots * 32); gid_in = A.take<uint32_t>(T); keys_in = A.take<uint64_t>(T);
        arrive = A.take<uint32_t>(NI); parent_int = A.take<int32_t>(NI); parent_leaf = A.take<int32_t>(T);
        tmp = A.take<char>(tmp_bytes ? tmp_bytes : 16);
        {   // everything that outlives the build: one allocation (and room for the traversal tree some builder will make over these leaves)
            const size_t sizes[15] = {(size_t)T * 4, (size_t)T * 8, (size_t)NI * 8, (size_t)NI * 12, (size_t)NI * 12, (size_t)T * 12, (size_t)T * 12, (size_t)T * sizeof(DevTri), (size_t)NI * sizeof(DevNode),
                                      (size_t)T * 4, (size_t)T * sizeof(DevShadeTri), 32, (size_t)NI * 8, (size_t)NI * 12, (size_t)NI * 12};
            size_t total = 0; for (size_t b : sizes) total += Arena::pad(b);
            HIPQ(hipMalloc(&out.block, total));
            char *p = out.block; auto cut = [&](size_t bytes) { char *q = p; p += Arena::pad(bytes); return q; };
            out.leaf_gid = (uint32_t *)cut(sizes[0]); out.keys = (uint64_t *)cut(sizes[1]); out.child = (int32_t *)cut(sizes[2]); out.node_lo = (float *)cut(sizes[3]); out.node_hi = (float *)cut(sizes[4]);
            out.leaf_lo = (float *)cut(sizes[5]); out.leaf_hi = (float *)cut(sizes[6]); out.tris = (DevTri *)cut(sizes[7]); out.nodes = (DevNode *)cut(sizes[8]); out.tri_prim = (uint32_t *)cut(sizes[9]);
            out.shade_tris = (DevShadeTri *)cut(sizes[10]); out.cbounds = (uint32_t *)cut(sizes[11]);
            out.res_trav_child = (int32_t *)cut(sizes[12]); out.res_trav_lo = (float *)cut(sizes[13]); out.res_trav_hi = (float *)cut(sizes[14]);
        }
        k_cb_init<<<1, kCbSlots, 0, s>>>(cslots);
        HIPQ(hipMemsetAsync(arrive, 0, (size_t)NI * 4, s));
        const uint32_t B = 256, GT = (T + B - 1) / B;
        k_soup<<<GT, B, 0, s>>>(in.prims, in.n_prims, in.prim_first_tri, T, triw, tlo, thi, out.tri_prim, cslots);
        k_cb_fold<<<1, kCbSlots, 0, s>>>(cslots, out.cbounds);
        k_morton<<<GT, B, 0, s>>>(T, in.morton_bits, tlo, thi, out.cbounds, keys_in, gid_in);
        HIPQ(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, out.keys, gid_in, out.leaf_gid, T, 0, 64, s));
        k_leaves<<<GT, B, 0, s>>>(T, out.leaf_gid, triw, tlo, thi, out.tri_prim, in.prim_first_tri, out.leaf_lo, out.leaf_hi, out.tris, in.prims, out.shade_tris);
        if (T > 1) {
            k_karras<<<(T - 1 + B - 1) / B, B, 0, s>>>((int)T, out.keys, out.leaf_gid, out.child, parent_int, parent_leaf);
            if (node_boxes) k_refit<<<GT, B, 0, s>>>(T, out.child, parent_int, parent_leaf, out.leaf_lo, out.leaf_hi, out.node_lo, out.node_hi, arrive); // (1.8 ms of arrival-counter waits on config 2: skipped when nothing will read the boxes)
        }
        if (node_boxes) k_emit_nodes<<<(NI + B - 1) / B, B, 0, s>>>(T, out.child, out.node_lo, out.node_hi, out.leaf_lo, out.leaf_hi, out.nodes);
        out.canon_boxes = node_boxes;
        HIPQ(hipGetLastError());
        HIPQ(hipStreamSynchronize(s));
        return hipSuccess;
    };
    err = body();
    own.release();
    if (err != hipSuccess) lbvh_free(out);
    return err;
}


void launch_emit_nodes(Lbvh &l, uint32_t T, hipStream_t s) { // traversal records from l.trav_child / trav_lo / trav_hi
    const uint32_t NI = T - 1;
    k_emit_nodes<<<(NI + 255) / 256, 256, 0, s>>>(T, l.trav_child, l.trav_lo, l.trav_hi, l.leaf_lo, l.leaf_hi, l.nodes);
}

// (Rounds 2 and 3 also built the traversal tree by parallel locally-ordered clustering over the Morton-ordered leaves -- ART_FLAG_DEVICE_TREE: 15 / 44 ms builds at 96-97 % of
// the SAH tree's ray rate -- until the binned SAH itself ran on the device in 3 / 11 ms (art_sahdev.hip); removed in round 4 with the other forms that lost.)

} // namespace art
