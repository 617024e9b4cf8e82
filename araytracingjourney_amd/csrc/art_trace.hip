// art_trace.hip -- per-frame wavefront pipeline (gfx950): primary rays + closest hit, hit reconstruction + PBR
// direct light + shadow-ray emission, shadow (any-hit) rays, accumulation.  Restates
// /root/reference/src/vk_renderer/shaders/rt_lightning_shadows/raytrace.rgen.glsl:77-200 (+ light.glsl, brdfs.glsl)
// split at its two traceRayEXT calls; the traversal replaces the driver/hardware behind traceRayEXT.
//
// Built with -ffp-contract=off: every fused multiply-add below is an explicit fmaf, so the geometry stages
// (ray generation, slab test, Moller-Trumbore, hit reconstruction up to the shadow ray) follow one fixed
// floating-point expression order.
//
// Geometry semantics (order- and structure-independent):
//   accept(tri, ray) := slab(AABB(tri), ray) passes AND Moller-Trumbore hits with tmin < t < tmax
//   t_eff := max(t_MT, t_entry(AABB(tri)));  closest := argmin (t_eff, gid);  any := exists accept
#include "art_internal.h"
#include <type_traits>

namespace art {

// ------------------------------------------------------------------------------------------------ vector helpers
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
    return mk(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ float len3(V3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ V3 nrm3(V3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return a * inv; }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }
__device__ __forceinline__ V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
// column-major mat4 times (x,y,z,w), rows 0..2: ((c0*x + c1*y) + c2*z) + c3*w
__device__ __forceinline__ V3 mat4_mul(const float *m, float x, float y, float z, float w) {
    return mk(((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w, ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w,
              ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w);
}
__device__ __forceinline__ V3 xform_point(const float *m, V3 p) {
    return mk(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
              ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]);
}
__device__ __forceinline__ V3 xform_vec(const float *m, V3 p) {
    return mk((m[0] * p.x + m[1] * p.y) + m[2] * p.z, (m[4] * p.x + m[5] * p.y) + m[6] * p.z, (m[8] * p.x + m[9] * p.y) + m[10] * p.z);
}

// ------------------------------------------------------------------------------------------------ traversal
struct Ray {
    V3 o, d;
    float tmin, tmax;
    V3 inv, ood;
};
__device__ __forceinline__ float safe_dir(float d) { return fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d; }
__device__ __forceinline__ void ray_init(Ray &r, V3 o, V3 d, float tmin, float tmax) {
    r.o = o; r.d = d; r.tmin = tmin; r.tmax = tmax;
    r.inv = mk(1.0f / safe_dir(d.x), 1.0f / safe_dir(d.y), 1.0f / safe_dir(d.z));
    r.ood = mk(o.x * r.inv.x, o.y * r.inv.y, o.z * r.inv.z);
}
__device__ __forceinline__ void ray_init_inv(Ray &r, V3 o, V3 d, V3 inv, float tmin, float tmax) { // inv = 1 / safe_dir(d), made elsewhere with the same operations
    r.o = o; r.d = d; r.tmin = tmin; r.tmax = tmax; r.inv = inv;
    r.ood = mk(o.x * inv.x, o.y * inv.y, o.z * inv.z);
}
// A ray with a non-finite origin or direction accepts no triangle (every Moeller-Trumbore quantity involves both, and a comparison with NaN is false),
// but it passes every box: fminf / fmaxf drop a NaN operand.  Its lane is switched off before the walk -- the same miss, without the walk of the whole tree.
// (0 * x is 0 for a finite x and NaN otherwise; nothing here is compiled with fast-math.)
__device__ __forceinline__ bool ray_finite(V3 o, V3 d) {
    float z = 0.0f * o.x; z = fmaf(0.0f, o.y, z); z = fmaf(0.0f, o.z, z); z = fmaf(0.0f, d.x, z); z = fmaf(0.0f, d.y, z); z = fmaf(0.0f, d.z, z);
    return z == 0.0f;
}
// monotone slab test against [tmin, tlimit]; tn = un-clamped entry distance
__device__ __forceinline__ bool slab(const Ray &r, float lx, float ly, float lz, float hx, float hy, float hz, float tlimit, float &tn) {
    float t0x = fmaf(lx, r.inv.x, -r.ood.x), t1x = fmaf(hx, r.inv.x, -r.ood.x);
    float t0y = fmaf(ly, r.inv.y, -r.ood.y), t1y = fmaf(hy, r.inv.y, -r.ood.y);
    float t0z = fmaf(lz, r.inv.z, -r.ood.z), t1z = fmaf(hz, r.inv.z, -r.ood.z);
    tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    return fmaxf(tn, r.tmin) <= fminf(tf, tlimit);
}
#define ART_BARY_EPS 1.0e-6f
// Moller-Trumbore, two-sided (instance flags 0, vk_model.rs:374), edges fattened by ART_BARY_EPS, tmin < t < tmax; e1 = v1 - v0, e2 = v2 - v0 (DevTri)
__device__ __forceinline__ bool moller_trumbore(const Ray &r, V3 v0, V3 e1, V3 e2, float &t, float &u, float &v) {
    V3 p = cross3(r.d, e2);
    float det = dot3(e1, p);
    if (det == 0.0f) return false;
    float inv = 1.0f / det;
    V3 tv = r.o - v0;
    float uu = dot3(tv, p) * inv;
    if (!(uu >= -ART_BARY_EPS && uu <= 1.0f + ART_BARY_EPS)) return false;
    V3 q = cross3(tv, e1);
    float vv = dot3(r.d, q) * inv;
    if (!(vv >= -ART_BARY_EPS && uu + vv <= 1.0f + ART_BARY_EPS)) return false;
    float tt = dot3(e2, q) * inv;
    if (!(tt > r.tmin && tt < r.tmax)) return false;
    t = tt; u = uu; v = vv;
    return true;
}

// the same arithmetic without the early exits: in a packet some lane nearly always survives each test, so the wave pays for
// every stage anyway and the exits only add exec-mask bookkeeping (det == 0 lanes compute inf/NaN that the flag discards)
__device__ __forceinline__ bool moller_trumbore_flat(const Ray &r, V3 v0, V3 e1, V3 e2, float &t, float &u, float &v) {
    V3 p = cross3(r.d, e2);
    float det = dot3(e1, p);
    float inv = 1.0f / det;
    V3 tv = r.o - v0;
    u = dot3(tv, p) * inv;
    V3 q = cross3(tv, e1);
    v = dot3(r.d, q) * inv;
    t = dot3(e2, q) * inv;
    return (det != 0.0f) & (u >= -ART_BARY_EPS) & (u <= 1.0f + ART_BARY_EPS) & (v >= -ART_BARY_EPS) & (u + v <= 1.0f + ART_BARY_EPS) & (t > r.tmin) & (t < r.tmax);
}

constexpr int kLdsStack = 16;   // per-lane short stack in LDS ([entry][lane], conflict-free); deeper entries spill to scratch
constexpr int kOvfStack = 80;   // 16 + 80 >= the deepest possible radix tree (63 key bits + 32 index bits)
constexpr int kBlock = 256;
constexpr int kFrameBlock = 64;  // the fused frame: one wave per workgroup
constexpr int kTraceBlock = 64;  // the persistent per-ray tracer: likewise (its waves share nothing either)
// tunables of the persistent tracer: {chunk, refill, blocks, leaf_batch}.  Measured on config 2 (profiles/README.md):
// one frame at a time is bound by the slowest wave's critical path -> small chunks, more waves; several frames in flight
// are throughput-bound -> fewer cursor atomics, fewer resident waves.  A context's ArtTuning (trace_chunk / trace_refill / trace_blocks / trace_leaf_batch)
// overrides the presets of ITS launches, for sweeps.
struct Tune { uint32_t chunk, refill, blocks, leaf_batch; };
// [0] / [1]: primary, shadow and query rays, one frame at a time / several in flight; [2] / [3]: AO rays likewise -- sixteen consecutive slots are one
// pixel's rays, a refill is cheap (k_ao_pixels + k_ao_table), and the kernel fits 8 waves per SIMD: larger chunks (a wave stays on 64 neighbouring
// pixels), all 8 192 wave slots (config 5: 13 700 -> 14 280 Mray/s over the presets of the other rays).  leaf_batch: lanes that must stand on a triangle
// before the wave runs the triangle test.  1 for the other rays (one frame of them at a time is bound by its slowest wave: waiting lanes lengthen it,
// round 1); AO launches are throughput-bound, and a triangle test run for every lone lane was a third of their issued instructions at a tenth of the
// lanes: 2 / 4 / 8 / 12 / 16 lanes -> 14 510 / 14 930 / 15 370 / 15 330 / 15 070 Mray/s on config 5
static const Tune kPreset[4] = {{64, 12, 1536, 1}, {128, 24, 1024, 1}, {256, 16, 2048, 8}, {1024, 24, 2048, 8}};
// the preset for a launch, with the context's overrides (they travel with the launch: nothing process-wide)
static Tune tune(bool pipelined, bool ao, const TraceTune &o) {
    Tune t = kPreset[(ao ? 2 : 0) + (pipelined ? 1 : 0)];
    if (o.chunk >= 64 && o.chunk <= 65536) t.chunk = o.chunk;
    if (o.refill >= 1 && o.refill <= 64) t.refill = o.refill;
    if (o.blocks >= 1 && o.blocks <= 16384) t.blocks = o.blocks;
    if (o.leaf_batch >= 1 && o.leaf_batch <= 64) t.leaf_batch = o.leaf_batch;
    return t;
}
constexpr int kCursorStride = 32; // one 128-byte line per XCD cursor

// local pixel id -> frame coordinates.  p = tile*1024 + sub*64 + lane; a wave covers an 8x8 pixel block.
__device__ __forceinline__ bool local_to_xy(uint32_t p, const uint32_t *__restrict__ tile_list, uint32_t tiles_x, uint32_t W, uint32_t H, uint32_t &x, uint32_t &y) {
    uint32_t tile = tile_list[p >> 10];
    uint32_t q = p & 1023u, sub = q >> 6, l = q & 63u;
    x = (tile % tiles_x) * kTile + (sub & 3u) * 8u + (l & 7u);
    y = (tile / tiles_x) * kTile + (sub >> 2) * 8u + (l >> 3);
    return x < W && y < H;
}

// ---- per-ray traversal state --------------------------------------------------------------------------------------
// `cur` and the stack hold child references: >= 0 an internal node of the structure being walked, < 0 a triangle
// (~leaf position).  Internal steps and triangle tests are separate so that the wave can batch the triangle tests: the
// Moeller-Trumbore block is ~150 instructions and, run whenever any single lane reaches a leaf, it was 60 % of all
// issued instructions at ~2 % lane utilisation.
constexpr int kOvfStack4 = 288; // 16 + 288 >= 3 pending siblings per level * 95 levels + 1
template <bool ANY, int OVF, int LDSN = kLdsStack> struct TravBase {   // LDSN: entries of the per-lane stack that live in LDS
    Ray r;
    float tbest, bu, bv;
    uint32_t bpos, bgid;
    int cur, sp;
    __device__ __forceinline__ void start(V3 o, V3 d, float tmin, float tmax) {
        ray_init(r, o, d, tmin, tmax);
        tbest = (ray_finite(o, d) && tmax == tmax) ? tmax : -INFINITY; // a non-finite ray (or range) fails the root's box test and ends as a miss
        bu = 0.f; bv = 0.f; bpos = kNoHit; bgid = kNoHit; cur = 0; sp = 0;
    }
    __device__ __forceinline__ void push(int ref, int *lds, int *ovf) {
        // the spill accesses are volatile so that the compiler keeps them apart from the LDS ones: merged, they become flat_load
        // / flat_store on a selected pointer, which waits on vmcnt AND lgkmcnt at every pop
        if (sp < LDSN) lds[sp * kTraceBlock] = ref; else if (sp < LDSN + OVF) *(volatile int *)&ovf[sp - LDSN] = ref;
        sp = min(sp + 1, LDSN + OVF); // the tree cannot need more (see the bounds above); never index past the spill area
    }
    __device__ __forceinline__ bool pop(int *lds, int *ovf) { // true: stack empty, the ray is finished
        if (sp == 0) return true;
        sp--;
        if (sp < LDSN) cur = lds[sp * kTraceBlock]; else cur = *(volatile int *)&ovf[sp - LDSN];
        return false;
    }
    // accept() of DESIGN.md 1.1 for the triangle in `cur`: exact triangle-AABB slab, then Moeller-Trumbore
    __device__ __forceinline__ bool step_leaf(const DevTri *__restrict__ tris, int *lds, int *ovf) {
        uint32_t pos = (uint32_t)~cur;
        // the reference of an ABSENT child of a 4-wide node (kAbsentChild: no leaf sits at position 2^31 - 1).  Its inverted box is left before it is entered by every
        // ray with a sign on every axis -- but a node scaled down to a point (a one-triangle tree, a node whose children were all masked) gives near == far on all three
        // axes to a ray through that point, which passes: such a "triangle" is nothing
        if (pos == 0x7FFFFFFFu) return pop(lds, ovf);
        const float4 *tq = reinterpret_cast<const float4 *>(tris + pos);
        float4 ta = tq[0], tb = tq[1], tc = tq[2], td = tq[3];   // v0 e1 e2 lo hi gid (DevTri)
        asm volatile("" : "+v"(td.x), "+v"(td.y), "+v"(td.z), "+v"(td.w), "+v"(tc.y), "+v"(tc.z), "+v"(tc.w)); // the box arrives with the vertices (else its load sinks below the triangle test: a second round trip for the lanes that pass it)
        float te, t, u, v;
        // triangle test first: the lane is here because this very box passed in the parent, so the slab (needed for t_eff and for
        // the conjunction) would nearly always run; after the triangle test it runs for the few hits only.  Same accept().
        if (moller_trumbore(r, mk(ta.x, ta.y, ta.z), mk(ta.w, tb.x, tb.y), mk(tb.z, tb.w, tc.x), t, u, v)) {
            if (slab(r, tc.y, tc.z, tc.w, td.x, td.y, td.z, tbest, te)) {
                if (ANY) { bpos = pos; tbest = t; return true; }
                float teff = fmaxf(t, te);
                uint32_t gid = __float_as_uint(td.w);
                if (teff < tbest || (teff == tbest && gid < bgid)) { tbest = teff; bu = u; bv = v; bpos = pos; bgid = gid; }
            }
        }
        return pop(lds, ovf);
    }
    // two children with hit flags / entry distances: continue with the nearer, stack the farther
    __device__ __forceinline__ bool descend2(bool h0, bool h1, float te0, float te1, int c0, int c1, int *lds, int *ovf) {
        if (h0 && h1) {
            bool first0 = te0 <= te1;
            push(first0 ? c1 : c0, lds, ovf);
            cur = first0 ? c0 : c1;
            return false;
        }
        if (h0) { cur = c0; return false; }
        if (h1) { cur = c1; return false; }
        return pop(lds, ovf);
    }
};

// 64-byte binary nodes (DevNode): both child boxes in full precision
template <bool ANY> struct Trav : TravBase<ANY, kOvfStack> {
    using Nodes = const DevNode *;
    __device__ __forceinline__ bool step_internal(Nodes nodes, int *lds, int *ovf) {
        const float4 *nq = reinterpret_cast<const float4 *>(nodes + this->cur);
        float4 q0 = nq[0], q1 = nq[1], q2 = nq[2], q3 = nq[3];
        float te0, te1;
        bool h0 = slab(this->r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, this->tbest, te0);
        bool h1 = slab(this->r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, this->tbest, te1);
        return this->descend2(h0, h1, te0, te1, __float_as_int(q3.x), __float_as_int(q3.y), lds, ovf);
    }
};

// 64-byte 4-wide quantised nodes (DevNode4)
template <bool ANY, int LDSN = kLdsStack> struct Trav4 : TravBase<ANY, kOvfStack4 + (kLdsStack - LDSN), LDSN> {
    using Nodes = const DevNode4 *;
    __device__ __forceinline__ bool step_internal(Nodes wide, int *lds, int *ovf) {
        const uint4 *nq = reinterpret_cast<const uint4 *>(wide + this->cur);
        uint4 a = nq[0], b = nq[1], c = nq[2], d = nq[3];
        // the child references arrive WITH the boxes: left to itself the compiler sinks their load below the box tests (it is only needed when a child is hit),
        // which makes every node step two dependent memory round trips instead of one -- and the AO launch waits on memory 44 % of the time (profiles round3):
        // config 5 16 980 -> 17 800 Mray/s
        asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
        float ox = __uint_as_float(a.x), oy = __uint_as_float(a.y), oz = __uint_as_float(a.z);
        float sx = __uint_as_float((a.w & 255u) << 23), sy = __uint_as_float(((a.w >> 8) & 255u) << 23), sz = __uint_as_float(((a.w >> 16) & 255u) << 23);
        int refs[4] = {(int)d.x, (int)d.y, (int)d.z, (int)d.w};
        float te[4]; bool h[4];
        // The slab with the entry plane of each axis chosen by the lane's direction sign -- ONE select per axis picks the word that holds the four children's near
        // planes, one the far planes -- instead of min / max of both planes per child: fma is monotone in the plane coordinate, so min(t0, t1) IS the near plane's
        // t for a box with lo <= hi, bit for bit (the packet walks' octant trick, per lane).  An absent child carries an inverted box (art_build.hip): whatever the
        // signs it is left before it is entered, so no valid-mask test.  33 -> 22 vector instructions a child; config 5 16 440 -> 16 980 Mray/s (profiles/README.md round 3).
        const Ray &r = this->r;
        const bool ngx = r.inv.x < 0.0f, ngy = r.inv.y < 0.0f, ngz = r.inv.z < 0.0f;
        const uint32_t nxw = ngx ? b.w : b.x, fxw = ngx ? b.x : b.w, nyw = ngy ? c.x : b.y, fyw = ngy ? b.y : c.x, nzw = ngz ? c.y : b.z, fzw = ngz ? b.z : c.y;
        const float lim = this->tbest;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float tnx = fmaf(fmaf((float)((nxw >> (8 * i)) & 255u), sx, ox), r.inv.x, -r.ood.x), tfx = fmaf(fmaf((float)((fxw >> (8 * i)) & 255u), sx, ox), r.inv.x, -r.ood.x);
            float tny = fmaf(fmaf((float)((nyw >> (8 * i)) & 255u), sy, oy), r.inv.y, -r.ood.y), tfy = fmaf(fmaf((float)((fyw >> (8 * i)) & 255u), sy, oy), r.inv.y, -r.ood.y);
            float tnz = fmaf(fmaf((float)((nzw >> (8 * i)) & 255u), sz, oz), r.inv.z, -r.ood.z), tfz = fmaf(fmaf((float)((fzw >> (8 * i)) & 255u), sz, oz), r.inv.z, -r.ood.z);
            float tn = fmaxf(fmaxf(tnx, tny), tnz), tf = fminf(fminf(tfx, tfy), tfz);
            te[i] = tn;
            h[i] = fmaxf(tn, r.tmin) <= fminf(tf, lim);
        }
        float tn = 3.0e38f; int ni = -1; // continue with the nearest hit child, stack the others
        if (ANY) { // an any-hit ray's answer does not depend on the order of its visits, only how soon a hit ends it: the first hit child in the node's own order (the
                   // children are sorted along an axis) instead of the nearest saves the selection chain -- config 5 15 650 -> 16 440 Mray/s (profiles/README.md round 3)
#pragma unroll
            for (int i = 3; i >= 0; i--) if (h[i]) ni = i;
        } else
#pragma unroll
        for (int i = 0; i < 4; i++) if (h[i] && te[i] < tn) { tn = te[i]; ni = i; }
        if (ni < 0) return this->pop(lds, ovf);
#pragma unroll
        for (int i = 0; i < 4; i++) if (h[i] && i != ni) this->push(refs[i], lds, ovf);
        this->cur = ni == 0 ? refs[0] : (ni == 1 ? refs[1] : (ni == 2 ? refs[2] : refs[3]));
        return false;
    }
};

// ---- packet traversal for primary rays -------------------------------------------------------------------------------
// The 64 rays of a wave are the pixels of one 8x8 block: almost the same path through the tree.  The wave walks ONE shared
// path (the union of its rays' paths): the node index is wave-uniform, so the node and triangle records come through the
// scalar cache instead of a 64-lane gather, there is no per-lane stack and no divergence outside the triangle test; each lane
// still tests its own ray against both child boxes with its own t_best.  A lane that would accept a triangle passes the slab
// test of every enclosing box (monotone slab), so the packet finds exactly the per-ray answer, bit for bit.
constexpr int kPacketStack = 288; // shared stack of node references per wave: >= 3 pending siblings per level * 95 levels // a shared stack of node references per wave; the radix tree is at most 95 levels deep
__device__ __forceinline__ bool ao_slot_decode(uint32_t slot, uint32_t spp, uint32_t &p, uint32_t &j);
__device__ __forceinline__ V3 ao_dir(V3 N, float tx, float ty, float tz);

__device__ __forceinline__ void ao_ray(const CameraArg &cam, uint32_t W, uint32_t H, uint32_t x, uint32_t y, float depth, float4 nm, uint32_t smp, V3 &o, V3 &d);

// v_cmp straight into an SGPR pair (HIP's __ballot(int) goes through v_cndmask + v_cmp_ne)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// streaming records (hits, shadow rays, contributions, frame outputs) are written once and read once: non-temporal accesses keep
// them from pushing the BVH out of the 4 MB L2s that the walks of all the frames in flight live in
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_nt(float4 *p, float4 v) { __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f4v *>(p)); }
__device__ __forceinline__ void st_nt(float *p, float v) { __builtin_nontemporal_store(v, p); }
// the compact tile buffer holds RGB only (12 B per pixel): the colour output's alpha is the constant 1 of imageStore(vec4(rho, 1)), so the
// gather moves three quarters of the bytes of the RGBA32F image and loses nothing
typedef float f3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void st_nt_rgb(float4 *tiles, size_t texel, float4 v) { __builtin_nontemporal_store(f3v{v.x, v.y, v.z}, reinterpret_cast<f3v *>(reinterpret_cast<float *>(tiles) + 3 * texel)); }
__device__ __forceinline__ float4 ld_nt(const float4 *p) { f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p)); return make_float4(v.x, v.y, v.z, v.w); }

// The walk of one packet: `cur` (node reference) and the stack are wave-uniform.  OCT 0..7: every ray of the packet has the
// direction signs (x: bit 0, y: bit 1, z: bit 2; set = negative) -- the entry plane of each axis is then known at compile time
// (fma is monotone in the plane coordinate, so min(t0, t1) IS the chosen one, bit for bit) and a box costs 6 fma + 4 min/max
// instead of 6 + 10.  OCT 8: mixed signs, the general slab.
template <int OCT> __device__ __forceinline__ bool slab_oct(const Ray &r, float lx, float ly, float lz, float hx, float hy, float hz, float tlimit, float &tn) {
    if (OCT >= 8) return slab(r, lx, ly, lz, hx, hy, hz, tlimit, tn);
    float nx = fmaf((OCT & 1) ? hx : lx, r.inv.x, -r.ood.x), fx = fmaf((OCT & 1) ? lx : hx, r.inv.x, -r.ood.x);
    float ny = fmaf((OCT & 2) ? hy : ly, r.inv.y, -r.ood.y), fy = fmaf((OCT & 2) ? ly : hy, r.inv.y, -r.ood.y);
    float nz = fmaf((OCT & 4) ? hz : lz, r.inv.z, -r.ood.z), fz = fmaf((OCT & 4) ? lz : hz, r.inv.z, -r.ood.z);
    tn = fmaxf(fmaxf(nx, ny), nz);
    float tf = fminf(fminf(fx, fy), fz);
    return fmaxf(tn, r.tmin) <= fminf(tf, tlimit);
}

// The packet walks read the tree through the CONSTANT address space: nothing writes nodes or triangles while a frame kernel runs, and only of
// constant-space loads does the compiler believe that after the frame's own stores (hits, depth, normal) -- a wave-uniform global load behind a store
// becomes a vector load of 64 equal addresses, which is what the shadow walks of k_frame were until round 2 (node data in 16 VGPRs, the vector L1's latency).
struct ConstQuads {
    typedef float Quad __attribute__((ext_vector_type(4)));
    __attribute__((address_space(4))) const Quad *p;
    __device__ __forceinline__ float4 operator[](int i) const { Quad v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
};
template <class T> __device__ __forceinline__ ConstQuads const_quads(const T *p) { ConstQuads q; q.p = (__attribute__((address_space(4))) const ConstQuads::Quad *)(uintptr_t)p; return q; }

// (Rounds 3 and 4 measured two uses of the packet's BEAM -- interval bounds of its rays against a node's four boxes on half a wave, 9 vector instructions -- and kept neither:
// as the node step itself its weaker culling bought 38 % more triangle steps (round 3b), as a FILTER in front of the per-ray box tests it left the launch's vector instructions
// where they were -- 124.98 M -> 124.86 M: what the filtered-out box tests save goes into the beam's set-up, its 9 instructions a step and the boxes it passes needlessly --
// and added 36 % scalar instructions and 13 % wave cycles: config 2 19 370 -> 17 250 Mray/s, frames bit-equal.  profiles/README.md round 4, profiles/round4_filter_*.)
#ifdef ART_PACKET_PROF
// profiling build only (make EXTRA=-DART_PACKET_PROF; tools/packet_prof.py): what the packet walks of k_frame are made of, summed over all waves since the last reset.
// [0..11] closest-hit (primary) walks, [12..23] any-hit (shadow) walks: walks, node steps, triangle steps, triangle steps that came off the stack, triangle steps in
// which some lane accepted the triangle, lanes that accepted, child boxes hit by some lane (of 4 per node step), mixed-octant walks
__device__ unsigned long long g_packet_prof[24];   // + [8] shader-clock cycles in node steps, [9] in triangle steps, [10] in the beam's set-up (each with the pop behind it)
#define PPROF(i, n) do { if ((threadIdx.x & 63u) == 0) atomicAdd(&g_packet_prof[(ANY ? 12 : 0) + (i)], (unsigned long long)(n)); } while (0)
#else
#define PPROF(i, n)
#endif
template <bool ANY, bool WIDE, int OCT, bool COUNT = false>
__device__ __forceinline__ void packet_walk(const FrameArgs &a, const Ray &r, bool &on, int *stk, float &tbest, float &bu, float &bv, uint32_t &bpos, uint32_t &bgid, uint32_t &steps) {
    int cur = 0, sp = 0; // wave-uniform
    constexpr int kPop = kAbsentChild;
#ifdef ART_PACKET_PROF
    unsigned long long pp_[10] = {1, 0, 0, 0, 0, 0, 0, OCT == 8, 0, 0}; bool from_stack_ = false;
#endif // "take the next node from the stack" (no leaf has position 2^31 - 1); also what an absent child of a 4-wide node refers to
    for (;;) {
#ifdef ART_PACKET_PROF
        const unsigned long long tk0_ = __builtin_amdgcn_s_memtime(); const bool was_node_ = cur >= 0;
#endif
        if (COUNT) steps++; // wave-uniform: nodes + triangles the packet visited (the fused frame's wave plan feeds on it; one s_add here costs 3.5 %, so only sampled frames count)
        if (cur >= 0 && WIDE) {
            // 4-wide node, float boxes (128 B, two scalar loads).  Its children were sorted at build time along the axis `ax` their centroids spread
            // most on, so the packet's front-to-back order is 0,1,2,3 or 3,2,1,0 by the sign of its rays' direction on that axis -- no per-lane
            // distances, no votes.  (Order only steers the culling: the answer is order-independent)  Absent children carry a
            // point box out at 3e38, which no ray passes (art_build.hip).  Measured against the binary walk (profiles/README.md r2): scalar
            // instructions -43 %, vector instructions +9 % (all four boxes of a node are tested, also below a child the binary walk would have
            // culled), half the dependent node fetches: config 2 +1 %, config 3 +8 %, config 4 +11 % rays/s.  The default since round 2
            // (ArtTuning.packet_wide = 2 selects the binary walk).
            ConstQuads nq = const_quads(a.widef + cur);
            float4 w0 = nq[0], w1 = nq[1], w2 = nq[2], w3 = nq[3], w4 = nq[4], w5 = nq[5], w6 = nq[6], w7 = nq[7];
            const int c0 = __float_as_int(w6.x), c1 = __float_as_int(w6.y), c2 = __float_as_int(w6.z), c3 = __float_as_int(w6.w);
            float te;
            const uint64_t m0 = ballot64(slab_oct<OCT>(r, w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, tbest, te));
            const uint64_t m1 = ballot64(slab_oct<OCT>(r, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, tbest, te));
            // (Round 4 also skipped the third and fourth box where a node has no such child -- its valid mask is a scalar, and a node of the collapse has three children on
            // average, so a quarter of the 52 box-test instructions of a step test a point box at 3e38: configs 2 / 3 / 4 19 450 / 30 990 / 17 080 Mray/s against 19 270-19 530 /
            // 31 450 / 17 170 -- nothing: the launch is not short of vector issue slots the way its 0.68 suggests; a step's chain of dependent scalar loads, ballots and the LDS
            // pop is what the waves take turns waiting for.  profiles/README.md round 4.)
            const uint64_t m2 = ballot64(slab_oct<OCT>(r, w3.x, w3.y, w3.z, w3.w, w4.x, w4.y, tbest, te));
            const uint64_t m3 = ballot64(slab_oct<OCT>(r, w4.z, w4.w, w5.x, w5.y, w5.z, w5.w, tbest, te));
#ifdef ART_PACKET_PROF
            pp_[1]++; pp_[6] += (m0 != 0ull) + (m1 != 0ull) + (m2 != 0ull) + (m3 != 0ull); from_stack_ = false;
#endif
            const uint32_t ax = __float_as_uint(w7.y);                   // 0..2 (wave-uniform)
            // mixed packets (OCT 8) go by the majority sign on that axis
            const bool rev = OCT < 8 ? ((OCT >> ax) & 1) != 0 : 2 * (int)__popcll(ballot64(on && (ax == 0 ? r.inv.x : (ax == 1 ? r.inv.y : r.inv.z)) < 0.0f)) > (int)__popcll(ballot64(on));
            const bool lane0 = (threadIdx.x & 63u) == 0;
            int next = kPop;
            if (!rev) { // near -> far = 0,1,2,3: stack the far ones first
                if (m3 != 0ull) next = c3;
                if (m2 != 0ull) { if (next != kPop) { if (lane0) stk[min(sp, kPacketStack - 1)] = next; sp = min(sp + 1, kPacketStack); } next = c2; }
                if (m1 != 0ull) { if (next != kPop) { if (lane0) stk[min(sp, kPacketStack - 1)] = next; sp = min(sp + 1, kPacketStack); } next = c1; }
                if (m0 != 0ull) { if (next != kPop) { if (lane0) stk[min(sp, kPacketStack - 1)] = next; sp = min(sp + 1, kPacketStack); } next = c0; }
            } else {
                if (m0 != 0ull) next = c0;
                if (m1 != 0ull) { if (next != kPop) { if (lane0) stk[min(sp, kPacketStack - 1)] = next; sp = min(sp + 1, kPacketStack); } next = c1; }
                if (m2 != 0ull) { if (next != kPop) { if (lane0) stk[min(sp, kPacketStack - 1)] = next; sp = min(sp + 1, kPacketStack); } next = c2; }
                if (m3 != 0ull) { if (next != kPop) { if (lane0) stk[min(sp, kPacketStack - 1)] = next; sp = min(sp + 1, kPacketStack); } next = c3; }
            }
            cur = next;
        } else if (cur >= 0) {
            ConstQuads nq = const_quads(a.nodes + cur);
            float4 q0 = nq[0], q1 = nq[1], q2 = nq[2], q3 = nq[3];
            int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
            float te0, te1;
            // lanes that are off carry tbest = -1: their slab tests fail by themselves, so each ballot is one v_cmp into an SGPR pair
            uint64_t m0 = ballot64(slab_oct<OCT>(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tbest, te0));
            uint64_t m1 = ballot64(slab_oct<OCT>(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tbest, te1));
            if (m0 != 0ull && m1 != 0ull) { // both: go where most rays enter first, stack the other
                uint64_t fl = ballot64(te0 <= te1), both = m0 & m1;
                uint64_t f0 = (m0 & ~m1) | (both & fl), f1 = (m1 & ~m0) | (both & ~fl);
                bool first0 = (int)__popcll(f0) >= (int)__popcll(f1);
                if ((threadIdx.x & 63u) == 0) stk[min(sp, kPacketStack - 1)] = first0 ? c1 : c0;
                sp = min(sp + 1, kPacketStack);
                cur = first0 ? c0 : c1;
            } else if (m0 != 0ull) cur = c0;
            else if (m1 != 0ull) cur = c1;
            else cur = kPop;
        } else {
            uint32_t pos = (uint32_t)~cur;
            ConstQuads tq = const_quads(a.tris + pos);
            float4 ta = tq[0], tb = tq[1], tc = tq[2], td = tq[3];   // v0 e1 e2 lo hi gid: one 64-byte scalar load (DevTri)
            float te = 0.f, t = 0.f, u = 0.f, v = 0.f;
            // accept() = slab(AABB(tri)) AND Moeller-Trumbore: the conjunction is evaluated triangle test first -- the parent already
            // tested this very box for the packet, so nearly every wave would pay for the slab, while few lanes survive the triangle test
            bool acc = moller_trumbore_flat(r, mk(ta.x, ta.y, ta.z), mk(ta.w, tb.x, tb.y), mk(tb.z, tb.w, tc.x), t, u, v) && on;
            if (acc) acc = slab_oct<OCT>(r, tc.y, tc.z, tc.w, td.x, td.y, td.z, tbest, te);
            // the ray state changes through selects, outside the divergent branches (no register copies around them)
            if (ANY) { bpos = acc ? pos : bpos; on = on && !acc; tbest = acc ? -1.0f : tbest; } // first accepted triangle: this lane is done
            else {
                float teff;   // = fmaxf(t, te): one v_max_f32 (fmaxf first quiets both operands, which are the results of arithmetic here: two more instructions a step)
                asm("v_max_f32 %0, %1, %2" : "=v"(teff) : "v"(t), "v"(te));
                uint32_t gid = __float_as_uint(td.w);
                bool better = acc & ((teff < tbest) | ((teff == tbest) & (gid < bgid))); // (bitwise: three compares and three mask operations, no nested exec regions)
                tbest = better ? teff : tbest; bu = better ? u : bu; bv = better ? v : bv; bpos = better ? pos : bpos; bgid = better ? gid : bgid;
            }
#ifdef ART_PACKET_PROF
            { uint64_t am_ = ballot64(acc); pp_[2]++; pp_[3] += from_stack_; pp_[4] += am_ != 0ull; pp_[5] += __popcll(am_); }
#endif
            cur = kPop;
            if (ANY && ballot64(on) == 0ull) break; // every ray of the packet is occluded
        }
        if (cur == kPop) {
            if (sp == 0) break;
            sp--;
            cur = __builtin_amdgcn_readfirstlane(stk[sp]); // same address in every lane: one broadcast LDS read (lane 0's write is ordered before it within the wave)
#ifdef ART_PACKET_PROF
            from_stack_ = true;
#endif
        }
#ifdef ART_PACKET_PROF
        pp_[was_node_ ? 8 : 9] += __builtin_amdgcn_s_memtime() - tk0_;
#endif
    }
#ifdef ART_PACKET_PROF
    for (int i = 0; i < 10; i++) PPROF(i, pp_[i]);
#endif
}

// one packet through the walk that fits its rays' direction signs
template <bool ANY, bool WIDE, bool COUNT = false>
__device__ __forceinline__ void walk_dispatch(const FrameArgs &a, const Ray &r, bool &on, int *stk, float &tbest, float &bu, float &bv, uint32_t &bpos, uint32_t &bgid, uint32_t &steps) {
    uint64_t act = ballot64(on);
    if (act == 0ull) return;
    // direction signs per axis: all set, none set, or mixed over the packet's rays
    uint64_t nx = ballot64(on && r.inv.x < 0.0f), ny = ballot64(on && r.inv.y < 0.0f), nz = ballot64(on && r.inv.z < 0.0f);
    bool uniform = (nx == 0ull || nx == act) && (ny == 0ull || ny == act) && (nz == 0ull || nz == act);
    int oct = !uniform ? 8 : (nx ? 1 : 0) | (ny ? 2 : 0) | (nz ? 4 : 0);
    switch (oct) {
    case 0: packet_walk<ANY, WIDE, 0, COUNT>(a, r, on, stk, tbest, bu, bv, bpos, bgid, steps); break;
    case 1: packet_walk<ANY, WIDE, 1, COUNT>(a, r, on, stk, tbest, bu, bv, bpos, bgid, steps); break;
    case 2: packet_walk<ANY, WIDE, 2, COUNT>(a, r, on, stk, tbest, bu, bv, bpos, bgid, steps); break;
    case 3: packet_walk<ANY, WIDE, 3, COUNT>(a, r, on, stk, tbest, bu, bv, bpos, bgid, steps); break;
    case 4: packet_walk<ANY, WIDE, 4, COUNT>(a, r, on, stk, tbest, bu, bv, bpos, bgid, steps); break;
    case 5: packet_walk<ANY, WIDE, 5, COUNT>(a, r, on, stk, tbest, bu, bv, bpos, bgid, steps); break;
    case 6: packet_walk<ANY, WIDE, 6, COUNT>(a, r, on, stk, tbest, bu, bv, bpos, bgid, steps); break;
    case 7: packet_walk<ANY, WIDE, 7, COUNT>(a, r, on, stk, tbest, bu, bv, bpos, bgid, steps); break;
    default: packet_walk<ANY, WIDE, 8, COUNT>(a, r, on, stk, tbest, bu, bv, bpos, bgid, steps); break;
    }
}

// ---- ray-traced ambient occlusion (BASELINE config 5): XeGTAO's I/O contract on the tracer ------------------------
// inputs: the frame's depth + view-space normal outputs (vk_xe_gtao.rs:295-333); noise: Hilbert index driving the R2
// sequence (main_pass.comp.hlsl:48-65, XeGTAO.h:120-142) with index + 288 * sample; cosine-weighted hemisphere.
__device__ __forceinline__ uint32_t hilbert_index(uint32_t x, uint32_t y) {
    uint32_t index = 0;
#pragma unroll
    for (uint32_t lvl = 32; lvl > 0; lvl /= 2) {
        uint32_t rx = (x & lvl) > 0, ry = (y & lvl) > 0;
        index += lvl * lvl * ((3u * rx) ^ ry);
        if (ry == 0) {
            if (rx == 1) { x = 63u - x; y = 63u - y; }
            uint32_t t = x; x = y; y = t;
        }
    }
    return index;
}
// cos/sin of u turns by quadrant reduction + Taylor polynomials in a fixed fmaf order (bit-reproducible, unlike sinf/cosf)
__device__ __forceinline__ void sincos_turns(float u, float &c, float &s) {
    float q = u * 4.0f;
    int k = (int)q;
    float x = (q - (float)k) * 1.57079632679489662f, x2 = x * x;
    float sp = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, -2.50521083854417188e-8f, 2.75573192239858907e-6f), -1.98412698412698413e-4f), 8.33333333333333333e-3f), -1.66666666666666667e-1f), 1.0f) * x;
    float cp = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 2.08767569878680990e-9f, -2.75573192239858907e-7f), 2.48015873015873016e-5f), -1.38888888888888889e-3f), 4.16666666666666667e-2f), -0.5f), 1.0f);
    k &= 3;
    c = k == 0 ? cp : (k == 1 ? -sp : (k == 2 ? -cp : sp));
    s = k == 0 ? sp : (k == 1 ? cp : (k == 2 ? -sp : -cp));
}
// an AO ray in three parts, so that the tracer's refill pays only for what differs from ray to ray: the pixel's point and world normal (once per
// pixel: k_ao_pixels), the sample's direction in the tangent frame (a function of the pixel's position in its 64x64 noise tile and the sample
// index only: a table, k_ao_table), and the frame itself.  ao_ray() composes the three: the same operations in the same order wherever a ray is made.
__device__ __forceinline__ void ao_pixel(const CameraArg &cam, uint32_t W, uint32_t H, uint32_t x, uint32_t y, float depth, float4 nm, V3 &o, V3 &N) {
    float px = (float)x + 0.5f, py = (float)y + 0.5f;
    float dx = (px / (float)W) * 2.0f - 1.0f, dy = (py / (float)H) * 2.0f - 1.0f;
    V3 org = mat4_mul(cam.view_inv, 0.f, 0.f, 0.f, 1.f);
    V3 tn = nrm3(mat4_mul(cam.proj_inv, dx, dy, 1.f, 1.f));
    V3 dir = mat4_mul(cam.view_inv, tn.x, tn.y, tn.z, 0.f);
    float sc = depth / -tn.z;
    o = mk(org.x + dir.x * sc, org.y + dir.y * sc, org.z + dir.z * sc); // the primary ray scaled to the stored view depth
    float nx = nm.x * 2.0f - 1.0f, ny = -(nm.y * 2.0f - 1.0f), nz = -(nm.z * 2.0f - 1.0f); // raytrace.rgen.glsl:192-194 inverted
    const float *VI = cam.view_inv;
    N = nrm3(mk((VI[0] * nx + VI[4] * ny) + VI[8] * nz, (VI[1] * nx + VI[5] * ny) + VI[9] * nz, (VI[2] * nx + VI[6] * ny) + VI[10] * nz));
}
// cosine-weighted direction of sample `smp` of the pixel whose Hilbert index in its 64x64 tile is `hil`, in the tangent frame: (r cos, r sin, sqrt(1 - u1))
__device__ __forceinline__ void ao_sample(uint32_t hil, uint32_t smp, float &tx, float &ty, float &tz) {
    float fi = (float)(hil + 288u * smp);
    float v1 = 0.5f + fi * 0.75487766624669276f, v2 = 0.5f + fi * 0.56984029099805327f;
    float u1 = v1 - floorf(v1), u2 = v2 - floorf(v2);
    float r = sqrtf(u1), cc, ss;
    tz = sqrtf(1.0f - u1);
    sincos_turns(u2, cc, ss);
    tx = r * cc; ty = r * ss;
}
__device__ __forceinline__ V3 ao_dir(V3 N, float tx, float ty, float tz) {
    float sg = copysignf(1.0f, N.z), aa = -1.0f / (sg + N.z), bb = N.x * N.y * aa; // branchless orthonormal basis (Duff et al. 2017)
    V3 T = mk(1.0f + sg * N.x * N.x * aa, sg * bb, -sg * N.x), B = mk(bb, sg + N.y * N.y * aa, -N.y);
    return (T * tx + B * ty) + N * tz;
}
__device__ __forceinline__ void ao_ray(const CameraArg &cam, uint32_t W, uint32_t H, uint32_t x, uint32_t y, float depth, float4 nm, uint32_t smp, V3 &o, V3 &d) {
    V3 N; float tx, ty, tz;
    ao_pixel(cam, W, H, x, y, depth, nm, o, N);
    ao_sample(hilbert_index(x & 63u, y & 63u), smp, tx, ty, tz);
    d = ao_dir(N, tx, ty, tz);
}
constexpr uint32_t kAoNoiseTile = 64 * 64; // the sample table has one entry per (sample, Hilbert index in the 64x64 tile)
// tab[j * 4096 + hil] = the tangent-frame direction of the pixel's j-th sample IN AZIMUTH ORDER (the R2 sequence's second coordinate, ties by sample index): a
// pixel's occlusion count is a sum over its samples, so their order is free -- and with it the tracer can put the samples of one azimuth quadrant of
// neighbouring pixels (similar normals, similar frames) into one wave: rays that leave in similar directions make similar walks (ao_slot_decode below).
__global__ __launch_bounds__(kBlock) void k_ao_table(uint32_t spp, float4 *__restrict__ tab) {
    uint32_t hil = blockIdx.x * kBlock + threadIdx.x;
    if (hil >= kAoNoiseTile) return;
    float key[64]; uint8_t idx[64];
    for (uint32_t s = 0; s < spp; s++) {
        float fi = (float)(hil + 288u * s), v2 = 0.5f + fi * 0.56984029099805327f;   // ao_sample's u2
        float k = v2 - floorf(v2);
        uint32_t at = s;
        while (at > 0 && key[at - 1] > k) { key[at] = key[at - 1]; idx[at] = idx[at - 1]; at--; }   // insertion keeps equal keys in sample order
        key[at] = k; idx[at] = (uint8_t)s;
    }
    for (uint32_t j = 0; j < spp; j++) {
        float tx, ty, tz;
        ao_sample(hil, idx[j], tx, ty, tz);
        tab[(size_t)j * kAoNoiseTile + hil] = make_float4(tx, ty, tz, 0.f);
    }
}
// The AO launch's slots.  Sixteen pixels -- one 4x4 quadrant of an 8x8 block -- own ceil(spp / 4) * 64 consecutive slots laid out
// [group of four samples in azimuth order][pixel][sample of the group]: the 64 slots a wave takes at a time are sixteen neighbouring pixels x the four samples of
// one azimuth quadrant.  (Round 2's order, [pixel][sample], put a pixel's whole hemisphere into sixteen neighbouring lanes: 55 % of the lanes of a vector
// instruction were active; this order: 61 %, 7 % fewer instructions.)  Slots past spp (spp not a multiple of four) are padding: no ray, occlusion byte 0.
__device__ __forceinline__ uint32_t ao_slots_per16(uint32_t spp) { return ((spp + 3u) >> 2) * 64u; }
__device__ __forceinline__ bool ao_slot_decode(uint32_t slot, uint32_t spp, uint32_t &p, uint32_t &j) {
    const uint32_t per16 = ao_slots_per16(spp), b16 = slot / per16, r = slot - b16 * per16;
    const uint32_t pi = (r >> 2) & 15u, q = b16 & 3u;
    const uint32_t x = (pi & 3u) + ((q & 1u) << 2), y = (pi >> 2) + ((q >> 1) << 2);
    p = (b16 >> 2) * 64u + y * 8u + x;    // local pixel: an 8x8 block is 64 consecutive local pixels, lane = y * 8 + x
    j = (r >> 6) * 4u + (r & 3u);
    return j < spp;
}
__device__ __forceinline__ size_t ao_slot_of(uint32_t p, uint32_t j, uint32_t spp) {
    const uint32_t lane = p & 63u, x = lane & 7u, y = lane >> 3, q = (x >> 2) | ((y >> 2) << 1), pi = (x & 3u) | ((y & 3u) << 2);
    return (size_t)((p >> 6) * 4u + q) * ao_slots_per16(spp) + (j >> 2) * 64u + pi * 4u + (j & 3u);
}

// AO rays are short: the 16 rays of a pixel stay inside a ball of the AO radius around one point.  Descend the 4-wide tree while
// exactly ONE child box overlaps that ball's bounding box -- every other subtree cannot hold a triangle the rays could reach --
// and let the pixel's rays start there (or skip them when nothing overlaps).  Quantised boxes contain the float boxes, so the test
// errs on the side of overlap; the rays' answers are those of a walk from the root (accept() is per triangle).
constexpr int kAoNothingNear = (int)0x80000000; // no leaf has position 2^31 - 1
// per local pixel, once: pix[2p] = the AO rays' origin | the node they start from (or kAoNothingNear: a miss pixel, or nothing within the radius),
// pix[2p + 1] = the world normal | the pixel's Hilbert index in its noise tile.  wide == null: every pixel starts at the root (binary walk, or entry search off).
__global__ __launch_bounds__(kBlock) void k_ao_pixels(FrameArgs a, const DevNode4 *__restrict__ wide, float radius, float4 *__restrict__ pix) {
    uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= a.n_local) return;
    uint32_t x, y;
    bool in = local_to_xy(p, a.tile_list, a.tiles_x, a.W, a.H, x, y);
    float depth = in ? a.depth[(size_t)y * a.W + x] : 10000.0f;
    if (!(depth < 10000.0f)) { pix[2 * (size_t)p] = make_float4(0.f, 0.f, 0.f, __int_as_float(kAoNothingNear)); pix[2 * (size_t)p + 1] = make_float4(0.f, 0.f, 1.f, 0.f); return; }
    V3 o, N;
    ao_pixel(a.cam, a.W, a.H, x, y, depth, a.normal[(size_t)y * a.W + x], o, N);
    int cur = 0;
    if (wide) {
        const float R = radius * 1.01f; // t runs to `radius` along a direction of length 1 +- rounding
        const float lox = o.x - R, loy = o.y - R, loz = o.z - R, hix = o.x + R, hiy = o.y + R, hiz = o.z + R;
        for (int level = 0; level < 64; level++) {
            const uint4 *nq = reinterpret_cast<const uint4 *>(wide + cur);
            uint4 qa = nq[0], qb = nq[1], qc = nq[2], qd = nq[3];
            float ox = __uint_as_float(qa.x), oy = __uint_as_float(qa.y), oz = __uint_as_float(qa.z);
            float sx = __uint_as_float((qa.w & 255u) << 23), sy = __uint_as_float(((qa.w >> 8) & 255u) << 23), sz = __uint_as_float(((qa.w >> 16) & 255u) << 23);
            uint32_t mask = qa.w >> 24;
            int refs[4] = {(int)qd.x, (int)qd.y, (int)qd.z, (int)qd.w};
            int n_over = 0, which = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float lx = fmaf((float)((qb.x >> (8 * i)) & 255u), sx, ox), ly = fmaf((float)((qb.y >> (8 * i)) & 255u), sy, oy), lz = fmaf((float)((qb.z >> (8 * i)) & 255u), sz, oz);
                float hx = fmaf((float)((qb.w >> (8 * i)) & 255u), sx, ox), hy = fmaf((float)((qc.x >> (8 * i)) & 255u), sy, oy), hz = fmaf((float)((qc.y >> (8 * i)) & 255u), sz, oz);
                bool over = ((mask >> i) & 1u) && lx <= hix && hx >= lox && ly <= hiy && hy >= loy && lz <= hiz && hz >= loz;
                if (over) { n_over++; which = refs[i]; }
            }
            if (n_over == 0) { cur = kAoNothingNear; break; }
            if (n_over > 1) break;          // the rays may go either way from here: this node is the entry
            cur = which;
            if (cur < 0) break;             // a single triangle is all there is
        }
    }
    pix[2 * (size_t)p] = make_float4(o.x, o.y, o.z, __int_as_float(cur));
    pix[2 * (size_t)p + 1] = make_float4(N.x, N.y, N.z, __uint_as_float(hilbert_index(x & 63u, y & 63u)));
}

// what a persistent tracing wave reads its rays from and writes its results to
enum { MODE_PRIMARY = 0, MODE_SHADOW = 1, MODE_QUERY_CLOSEST = 2, MODE_QUERY_ANY = 3, MODE_AO = 4 };
struct TraceArgs {
    const DevNode *nodes; const DevNode4 *wide; const DevTri *tris;
    uint32_t total;          // candidate slots
    uint32_t leaf_batch;     // lanes that must wait on a triangle before the wave runs the triangle test
    uint32_t chunk, refill;  // slots a wave takes from a cursor at a time; idle lanes that trigger a refill (Aila & Laine 2009, dynamic fetch)
    uint32_t *cursors;       // 8 per-XCD chunk cursors, kCursorStride words apart (zeroed before the launch)
    uint32_t *count;         // rays actually traced (MODE_SHADOW), may be null
    // MODE_PRIMARY
    CameraArg cam; uint32_t W, H; const uint32_t *tile_list; uint32_t tiles_x;
    float4 *hits;
    // MODE_SHADOW / MODE_QUERY_*: rays[2*slot] = o.xyz,tmax(<=0: no ray) | rays[2*slot+1] = d.xyz,tmin (queries) / unused
    const float4 *rays;
    float4 *contrib; uint32_t n_local; uint32_t *shadow_bits;
    uint32_t *any_out;
    // MODE_AO: rays are generated from the frame's depth + view-space normal outputs (XeGTAO's inputs); slot -> (local pixel, sample): ao_slot_decode
    const float *depth; const float4 *normal; uint32_t spp; float ao_radius; uint8_t *occl;
    const float4 *ao_pix;     // MODE_AO: per local pixel, origin | start node and world normal | Hilbert index (k_ao_pixels)
    const float4 *ao_tab;     // MODE_AO: [sample][Hilbert index] tangent-frame direction (k_ao_table)
};

#ifdef ART_TRACE_PROF
// profiling build only (make EXTRA=-DART_TRACE_PROF; tools/trace_prof.py): what the iterations of the persistent tracer's waves are made of, summed over the waves of every launch since
// the last reset: [0] loop iterations, [1] iterations with a node step, [2] lanes in them, [3] iterations with a triangle step, [4] lanes in them, [5] iterations with a refill, [6] lanes
// refilled (= rays), [7] lanes with a ray summed over all iterations
__device__ unsigned long long g_trace_prof[8];
#define TPROF(i, n) prof_[i] += (n)
#else
#define TPROF(i, n)
#endif
// Persistent-threads wavefront tracer.  Each wave keeps up to 64 rays in flight; when kRefill or more lanes have
// finished it compacts the idle lanes with __ballot / mbcnt and hands them the next candidates of its chunk; chunks
// come from eight per-XCD work cursors (one returning atomic per chunk), so neighbouring rays stay on one XCD's L2.
// Every wave exits once all cursors are exhausted and its lanes are idle.
template <int MODE, int WIDTH>
__global__ __launch_bounds__(kTraceBlock) __attribute__((amdgpu_waves_per_eu(MODE == MODE_AO ? 8 : 4, 8))) void k_trace(TraceArgs a) {
    constexpr bool ANY = MODE == MODE_SHADOW || MODE == MODE_QUERY_ANY || MODE == MODE_AO;
    __shared__ int stack[kLdsStack * kTraceBlock];
    int ovf[WIDTH == 4 ? kOvfStack4 : kOvfStack];
    const uint32_t leaf_batch = a.leaf_batch;
    int *lds = &stack[threadIdx.x];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_chunks = (a.total + a.chunk - 1) / a.chunk; // <= 2^25, so n_chunks * 8 fits

    uint32_t shard = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; // HW_REG_XCC_ID: speed only
    uint32_t shards_left = 8;
    uint32_t cur = 0, end = 0; // wave-uniform: the unread part of this wave's chunk
    bool exhausted = false, active = false;
    typename std::conditional<WIDTH == 4, Trav4<ANY>, Trav<ANY>>::type tr;
    uint32_t slot = 0, traced = 0;
#ifdef ART_TRACE_PROF
    unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        uint64_t idle = __ballot(!active);
        TPROF(0, 1); TPROF(7, 64 - __popcll(idle));
        uint32_t n_idle = (uint32_t)__popcll(idle);
        if (!exhausted && n_idle >= a.refill) {
            if (cur == end) { // take the next chunk: lane 0 pops, everyone learns the result
                uint32_t got = 0xFFFFFFFFu;
                if (lane == 0) {
                    while (shards_left) { // shard s owns chunks [n*s/8, n*(s+1)/8): contiguous, so an XCD keeps a screen region
                        uint32_t lo = (n_chunks * shard) >> 3, hi = (n_chunks * (shard + 1u)) >> 3;
                        uint32_t c = hi > lo ? atomicAdd(&a.cursors[shard * kCursorStride], 1u) : 0u;
                        if (hi > lo && c < hi - lo) { got = lo + c; break; }
                        shard = (shard + 1u) & 7u; shards_left--;
                    }
                }
                got = __builtin_amdgcn_readfirstlane(got);
                shard = __builtin_amdgcn_readfirstlane(shard);
                shards_left = __builtin_amdgcn_readfirstlane(shards_left);
                if (got >= n_chunks) exhausted = true; // (also the never-expected out-of-range pop: no slot beyond total is touched)
                else { cur = got * a.chunk; end = min(cur + a.chunk, a.total); }
            }
            if (!exhausted) {
                uint32_t avail = end - cur;
                uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (!active && rank < avail) {
                    uint32_t sidx = cur + rank;
                    if (MODE == MODE_PRIMARY) {
                        uint32_t x, y;
                        if (local_to_xy(sidx, a.tile_list, a.tiles_x, a.W, a.H, x, y)) { // raytrace.rgen.glsl:78-88
                            float px = (float)x + 0.5f, py = (float)y + 0.5f;
                            float ux = px / (float)a.W, uy = py / (float)a.H;
                            float dx = ux * 2.0f - 1.0f, dy = uy * 2.0f - 1.0f;
                            V3 org = mat4_mul(a.cam.view_inv, 0.f, 0.f, 0.f, 1.f);
                            V3 tgt = nrm3(mat4_mul(a.cam.proj_inv, dx, dy, 1.f, 1.f));
                            V3 dir = mat4_mul(a.cam.view_inv, tgt.x, tgt.y, tgt.z, 0.f);
                            tr.start(org, dir, 0.001f, 10000.0f);
                            active = true;
                        } else a.hits[sidx] = make_float4(10000.0f, 0.f, 0.f, __uint_as_float(kNoHit));
                    } else if (MODE == MODE_AO) {
                        // everything but the direction was made once per pixel (k_ao_pixels), the direction's tangent-frame part once per context (k_ao_table):
                        // a refill is three 16-byte loads, a frame from the normal and ray_init -- cheap enough to refill at few idle lanes
                        uint32_t p, smp;                               // smp: the sample's rank in the pixel's azimuth order (k_ao_table)
                        const bool real = ao_slot_decode(sidx, a.spp, p, smp);
                        float4 po = real ? a.ao_pix[2 * (size_t)p] : make_float4(0.f, 0.f, 0.f, __int_as_float(kAoNothingNear));
                        float4 pn = real ? a.ao_pix[2 * (size_t)p + 1] : make_float4(0.f, 0.f, 1.f, 0.f);   // both halves of the pixel record in one round trip (the table entry needs the second)
                        asm volatile("" : "+v"(pn.x), "+v"(pn.y), "+v"(pn.z), "+v"(pn.w));
                        int entry = __float_as_int(po.w);
                        if (entry == kAoNothingNear) a.occl[sidx] = 0; // a padding slot, a miss pixel, or no box within the AO radius: unoccluded, nothing to trace
                        else {
                            float4 t = a.ao_tab[smp * kAoNoiseTile + __float_as_uint(pn.w)];
                            tr.start(mk(po.x, po.y, po.z), ao_dir(mk(pn.x, pn.y, pn.z), t.x, t.y, t.z), a.ao_radius * 0.01f, a.ao_radius);
                            tr.cur = entry;                            // the walk starts below the part of the tree that every ray of this pixel would cross alike
                            active = true;
                        }
                    } else {
                        float4 r0 = a.rays[2 * (size_t)sidx];
                        if (MODE != MODE_SHADOW || r0.w > 0.0f) {
                            float4 r1 = a.rays[2 * (size_t)sidx + 1];
                            // shadow rays: tmin 0.01 (raytrace.rgen.glsl:174); queries carry their own tmin
                            if (MODE == MODE_SHADOW) tr.start(mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z), 0.01f, r0.w);
                            else tr.start(mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z), r0.w, r1.w);
                            active = true;
                        }
                    }
                    if (active) { slot = sidx; traced++; }
                }
                TPROF(5, 1); TPROF(6, __popcll(__ballot(active) & idle));
                cur += min(n_idle, avail);
            }
            if (__ballot(active) == 0ull) continue; // nothing to trace yet: fetch again (or find the cursors exhausted)
        } else if (n_idle == 64u) {
            if (exhausted) break;
            continue;
        }
        // one internal step for every lane standing on a node ...
        bool done = false;
        { uint64_t nm_ = __ballot(active && tr.cur >= 0); if (nm_) { TPROF(1, 1); TPROF(2, __popcll(nm_)); } }
#ifndef ART_NODE_REPS
#define ART_NODE_REPS 3   // node steps a lane may take per turn of the loop (the turn's ballots, refill test and branches are scalar work the wave pays per turn: config 5 19 000 -> 20 000 Mray/s at 3, 19 900 at 2, 19 000 at 6)
#endif
#pragma unroll
        for (int rep_ = 0; rep_ < ART_NODE_REPS; rep_++)
        if (active && !done && tr.cur >= 0) {
            if constexpr (WIDTH == 4) done = tr.step_internal(a.wide, lds, ovf);
            else done = tr.step_internal(a.nodes, lds, ovf);
        }
        // ... and the triangle tests only once enough lanes wait on one (or nobody can move without it)
        bool on_leaf = active && !done && tr.cur < 0;
        uint64_t lm = __ballot(on_leaf);
        if (lm != 0ull && ((uint32_t)__popcll(lm) >= leaf_batch || __ballot(active && !done && tr.cur >= 0) == 0ull)) {
            TPROF(3, 1); TPROF(4, __popcll(lm));
            if (on_leaf) done = tr.step_leaf(a.tris, lds, ovf);
        }
        if (done) {
            active = false;
            if (MODE == MODE_PRIMARY || MODE == MODE_QUERY_CLOSEST)
                a.hits[slot] = tr.bpos != kNoHit ? make_float4(tr.tbest, tr.bu, tr.bv, __uint_as_float(tr.bpos))
                                                 : make_float4(MODE == MODE_PRIMARY ? 10000.0f : tr.r.tmax, 0.f, 0.f, __uint_as_float(kNoHit));
            else if (MODE == MODE_QUERY_ANY) a.any_out[slot] = tr.bpos != kNoHit ? 1u : 0u;
            else if (MODE == MODE_AO) a.occl[slot] = tr.bpos != kNoHit ? 1 : 0;
            else if (tr.bpos != kNoHit) { // shadowed: the light keeps 0.05 of its contribution (raytrace.rgen.glsl:179-181)
                float4 c = a.contrib[slot];
                a.contrib[slot] = make_float4(c.x * 0.05f, c.y * 0.05f, c.z * 0.05f, c.w);
                if (a.shadow_bits) { uint32_t i = slot / a.n_local; if (i < 16) atomicOr(&a.shadow_bits[slot - i * a.n_local], 1u << i); }
            }
        }
    }
#ifdef ART_TRACE_PROF
    if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&g_trace_prof[i], prof_[i]);
#endif
    if (MODE == MODE_SHADOW && a.count) { // rays this wave traced: one atomic per wave
        for (int off = 32; off >= 1; off >>= 1) traced += (uint32_t)__shfl_xor((int)traced, off);
        if (lane == 0 && traced) atomicAdd(a.count + (blockIdx.x % kSlotCount) * kSlotStride, traced);   // one wave per workgroup
    }
}

// The AO launch's own persistent tracer (round 4): rays are MADE by the whole wave, sixty-four at a time, into a pool in LDS, and a lane that finishes TAKES its next ray from
// the pool at once (a dozen LDS reads) -- in k_trace a finished lane waits until two dozen lanes are idle, because a refill there is ~80 instructions whoever runs it: 48 of
// 64 lanes held a ray (profiles/README.md round 3).  Same slots, same rays, same walks (Trav4, any hit): the occlusion bytes cannot change.
constexpr int kAoLds = 8;           // per-lane stack entries in LDS (AO rays start deep in the tree and are short: the walk rarely holds more; the rest spills)
constexpr int kAoPoolFields = 11;   // o.xyz d.xyz inv.xyz entry slot
constexpr uint32_t kAoPoolTake = 8; // idle lanes at which the wave turns to the pool (ArtTuning.trace_refill overrides)
__global__ __launch_bounds__(kTraceBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_trace_ao(TraceArgs a) {
    __shared__ int stack[kAoLds * kTraceBlock];
    __shared__ float pool[kAoPoolFields][kTraceBlock];
    int ovf[kOvfStack4 + (kLdsStack - kAoLds)];
    const uint32_t leaf_batch = a.leaf_batch;
    int *lds = &stack[threadIdx.x];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_chunks = (a.total + a.chunk - 1) / a.chunk;
    uint32_t shard = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; // HW_REG_XCC_ID: speed only
    uint32_t shards_left = 8;
    uint32_t cur = 0, end = 0;   // wave-uniform: the unread part of this wave's chunk
    uint32_t pool_n = 0;         // wave-uniform: rays in the pool
    bool exhausted = false, active = false;
    Trav4<true, kAoLds> tr;
    tr.r.tmin = a.ao_radius * 0.01f; tr.r.tmax = a.ao_radius;
    uint32_t slot = 0;
    for (;;) {
        const uint64_t idle = ballot64(!active);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle >= a.refill) {
            if (pool_n == 0 && !exhausted) {      // the pool is empty: the WHOLE wave makes the next (up to) 64 rays of its chunk
                if (cur == end) {
                    uint32_t got = 0xFFFFFFFFu;
                    if (lane == 0) {
                        while (shards_left) {
                            uint32_t lo = (n_chunks * shard) >> 3, hi = (n_chunks * (shard + 1u)) >> 3;
                            uint32_t c = hi > lo ? atomicAdd(&a.cursors[shard * kCursorStride], 1u) : 0u;
                            if (hi > lo && c < hi - lo) { got = lo + c; break; }
                            shard = (shard + 1u) & 7u; shards_left--;
                        }
                    }
                    got = __builtin_amdgcn_readfirstlane(got);
                    shard = __builtin_amdgcn_readfirstlane(shard);
                    shards_left = __builtin_amdgcn_readfirstlane(shards_left);
                    if (got >= n_chunks) exhausted = true;
                    else { cur = got * a.chunk; end = min(cur + a.chunk, a.total); }
                }
                if (!exhausted) {
                    const uint32_t take = min(64u, end - cur), sidx = cur + lane;
                    bool has = false;
                    V3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 1.f); int entry = kAoNothingNear;
                    if (lane < take) {
                        uint32_t p, smp;
                        const bool real = ao_slot_decode(sidx, a.spp, p, smp);
                        float4 po = real ? a.ao_pix[2 * (size_t)p] : make_float4(0.f, 0.f, 0.f, __int_as_float(kAoNothingNear));
                        float4 pn = real ? a.ao_pix[2 * (size_t)p + 1] : make_float4(0.f, 0.f, 1.f, 0.f);
                        asm volatile("" : "+v"(pn.x), "+v"(pn.y), "+v"(pn.z), "+v"(pn.w));
                        entry = __float_as_int(po.w);
                        if (entry != kAoNothingNear) {
                            float4 t = a.ao_tab[smp * kAoNoiseTile + __float_as_uint(pn.w)];
                            o = mk(po.x, po.y, po.z); d = ao_dir(mk(pn.x, pn.y, pn.z), t.x, t.y, t.z);
                            has = ray_finite(o, d);   // (a non-finite ray accepts nothing: unoccluded, like tr.start's dead ray)
                        }
                        if (!has) a.occl[sidx] = 0;    // a padding slot, a miss pixel, no box within the AO radius, a dead ray: nothing to trace
                    }
                    const uint64_t hm = ballot64(has);
                    if (has) {
                        const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                        const V3 inv = mk(1.0f / safe_dir(d.x), 1.0f / safe_dir(d.y), 1.0f / safe_dir(d.z));   // ray_init's operations
                        pool[0][at] = o.x; pool[1][at] = o.y; pool[2][at] = o.z; pool[3][at] = d.x; pool[4][at] = d.y; pool[5][at] = d.z;
                        pool[6][at] = inv.x; pool[7][at] = inv.y; pool[8][at] = inv.z; pool[9][at] = __int_as_float(entry); pool[10][at] = __uint_as_float(sidx);
                    }
                    pool_n = (uint32_t)__popcll(hm);
                    cur += take;
                }
            }
            if (pool_n) {                             // idle lanes take rays from the top of the pool
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (!active && rank < pool_n) {
                    const uint32_t at = pool_n - 1u - rank;
                    tr.r.o = mk(pool[0][at], pool[1][at], pool[2][at]); tr.r.d = mk(pool[3][at], pool[4][at], pool[5][at]); tr.r.inv = mk(pool[6][at], pool[7][at], pool[8][at]);
                    tr.r.ood = mk(tr.r.o.x * tr.r.inv.x, tr.r.o.y * tr.r.inv.y, tr.r.o.z * tr.r.inv.z);
                    tr.tbest = tr.r.tmax; tr.bu = 0.f; tr.bv = 0.f; tr.bpos = kNoHit; tr.bgid = kNoHit; tr.sp = 0;
                    tr.cur = __float_as_int(pool[9][at]); slot = __float_as_uint(pool[10][at]);
                    active = true;
                }
                pool_n -= min(n_idle, pool_n);
            } else if (exhausted) { if (n_idle == 64u) break; }
            if (ballot64(active) == 0ull) continue;   // nothing to trace yet (a chunk of padding / misses): make more
        }
        bool done = false;
#pragma unroll
        for (int rep_ = 0; rep_ < ART_NODE_REPS; rep_++)
            if (active && !done && tr.cur >= 0) done = tr.step_internal(a.wide, lds, ovf);
        const bool on_leaf = active && !done && tr.cur < 0;
        const uint64_t lm = ballot64(on_leaf);
        if (lm != 0ull && ((uint32_t)__popcll(lm) >= leaf_batch || ballot64(active && !done && tr.cur >= 0) == 0ull)) { if (on_leaf) done = tr.step_leaf(a.tris, lds, ovf); }
        if (done) { active = false; a.occl[slot] = tr.bpos != kNoHit ? 1 : 0; }
    }
}

// ------------------------------------------------------------------------------------------------ lights (light.glsl)
__device__ V3 compute_barycentric(V3 a, V3 b, V3 c, V3 p) { // light.glsl:50-68
    V3 v0 = b - a, v1 = c - a, v2 = p - a;
    float d00 = dot3(v0, v0), d01 = dot3(v0, v1), d11 = dot3(v1, v1), d20 = dot3(v2, v0), d21 = dot3(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    V3 r;
    r.x = (d11 * d20 - d01 * d21) / denom;
    r.y = (d00 * d21 - d01 * d20) / denom;
    r.z = 1.0f - r.x - r.y;
    return r;
}
__device__ V3 closest_point_to_segment(V3 p0, V3 p1, V3 p) { // light.glsl:70-75
    V3 v01 = p1 - p0;
    float t = dot3(p - p0, v01) / dot3(v01, v01);
    t = clampf(t, 0.0f, 1.0f);
    return p0 + v01 * t;
}
__device__ V3 closest_point_to_triangle(V3 p0, V3 p1, V3 p2, V3 pt) { // light.glsl:77-91
    V3 b = compute_barycentric(p0, p1, p2, pt);
    if (b.x < 0.0f) return closest_point_to_segment(p2, p0, pt);
    else if (b.z < 0.0f) return closest_point_to_segment(p1, p2, pt);
    return pt;
}
__device__ V3 get_unnormalized_L_vec(const ArtLight &l, V3 pos) { // light.glsl:93-124
    if (l.type == 0u || l.type == 1u) return ld3(l.pos) - pos;
    if (l.type == 2u) return neg(ld3(l.dir)) * 10.0f;
    if (l.type == 3u) {
        V3 ldir = ld3(l.dir), lp = ld3(l.pos), p2 = ld3(l.area_pos2), p3 = ld3(l.area_pos3);
        float distance = dot3(ldir, p2) - dot3(ldir, pos);
        V3 cp = pos + ldir * distance;
        V3 b = compute_barycentric(lp, p2, p3, cp);
        V3 c;
        if (b.x < 0.0f) { V3 p4 = (lp - p2) + p3; c = closest_point_to_triangle(lp, p3, p4, cp); }
        else if (b.y < 0.0f) c = closest_point_to_segment(lp, p2, cp);
        else if (b.z < 0.0f) c = closest_point_to_segment(p2, p3, cp);
        else c = cp;
        return c - pos;
    }
    return mk(1.0f, 1.0f, 1.0f);
}
__device__ V3 get_light_radiance(const ArtLight &l, V3 pos, V3 L) { // light.glsl:34-48
    V3 rad = ld3(l.color);
    if (l.type == 1u || l.type == 3u) {
        float theta_s = acosf(clampf(dot3(ld3(l.dir), neg(L)), -1.0f, 1.0f));
        float t = clampf((theta_s - l.umbra_angle) / (l.penumbra_angle - l.umbra_angle), 0.0f, 1.0f);
        rad = rad * (t * t);
    }
    if (l.falloff_distance > 0.0f) {
        float q = __fdividef(len3(ld3(l.pos) - pos), l.falloff_distance);
        float w = fmaxf(1.0f - q * q, 0.0f);
        rad = rad * (w * w);
    }
    return rad;
}

// ------------------------------------------------------------------------------------------------ BRDFs (brdfs.glsl)
#define ART_INV_PI (1.0f / 3.14159265359f)
__device__ __forceinline__ float D_GGX(float a_, float NdotH) { // brdfs.glsl:6-14
    float om = 1.0f - NdotH * NdotH;
    float a = NdotH * a_;
    float k = __fdividef(a_, om + a * a);
    return k * k * ART_INV_PI;
}
__device__ __forceinline__ float V_SmithGGXCorrelated_fast(float a_, float NdotV, float NdotL) { // brdfs.glsl:25-29
    return __fdividef(0.5f, mixf(2.0f * NdotL * NdotV, NdotL + NdotV, a_));
}
__device__ __forceinline__ float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
__device__ __forceinline__ float F_Schlick1(float F0, float F90, float x) { return F0 + (F90 - F0) * pow5(1.0f - x); } // brdfs.glsl:44-49
__device__ float Burley_diffuse_local_sss(float a_, float NdotV, float nc_NdotV, float nc_NdotL, float LdotH, float ratio) { // brdfs.glsl:89-99
    float F_SS90 = a_ * LdotH * LdotH;
    float F_SS = F_Schlick1(1.0f, F_SS90, nc_NdotL) * F_Schlick1(1.0f, F_SS90, nc_NdotV);
    float f_ss = (__fdividef(1.0f, nc_NdotV * nc_NdotL) - 0.5f) * F_SS + 0.5f;
    float local_sss = 1.25f * ratio * f_ss;
    float f90 = 0.5f + 2.0f * F_SS90;
    float diffuse = (1.0f - ratio) * F_Schlick1(1.0f, f90, nc_NdotL) * F_Schlick1(1.0f, f90, nc_NdotV);
    return NdotV * (diffuse + local_sss) * ART_INV_PI;
}

// ------------------------------------------------------------------------------------------------ textures
// sampler2DArray, linear / REPEAT, LOD 0 (no derivatives in a raygen stage): vk_rt_descriptor_set.rs:42-56
__device__ __forceinline__ int wrapi(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
__device__ float4 sample_tex(const uint32_t *__restrict__ pool, const DevPrim &P, int layer, float u, float v) {
    int tw = (int)P.tw, th = (int)P.th;
    float x = u * (float)tw - 0.5f, y = v * (float)th - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = wrapi((int)x0f, tw), y0 = wrapi((int)y0f, th);
    int x1 = wrapi(x0 + 1, tw), y1 = wrapi(y0 + 1, th);
    const uint32_t *base = pool + P.texture_offset + (size_t)layer * tw * th;
    uint32_t t00 = base[(size_t)y0 * tw + x0], t10 = base[(size_t)y0 * tw + x1], t01 = base[(size_t)y1 * tw + x0], t11 = base[(size_t)y1 * tw + x1];
    const float k = 1.0f / 255.0f;
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float A = (float)((t00 >> (8 * c)) & 255u) * k, B = (float)((t10 >> (8 * c)) & 255u) * k;
        float Cc = (float)((t01 >> (8 * c)) & 255u) * k, D = (float)((t11 >> (8 * c)) & 255u) * k;
        float top = A * (1.0f - fx) + B * fx, bot = Cc * (1.0f - fx) + D * fx;
        o[c] = top * (1.0f - fy) + bot * fy;
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

// ---- shading (raytrace.rgen.glsl:103-199), shared by the staged frame (k_shade) and the fused frame (k_frame) ---------
struct Surface { V3 world_pos, N, Vv, albedo; float metallic, alpha, nc_NdotV, NdotV; };

// rgen:107-150: the hit triangle's attributes, normal mapping, material; also the frame's depth / view-space normal outputs
__device__ __forceinline__ void shade_surface(const FrameArgs &a, const CameraArg &cam, uint32_t pos, float hu, float hv, Surface &S, float &out_depth, V3 &out_normal) {
    // one dependent fetch: the shading record holds what get_indices + three vertex reads would return (rgen:107-114)
    const float4 *sq = reinterpret_cast<const float4 *>(a.shade_tris + pos);
    float4 s0 = sq[0], s1 = sq[1], s2 = sq[2], s3 = sq[3], s4 = sq[4], s5 = sq[5], s6 = sq[6], s7 = sq[7], s8 = sq[8];
    const DevPrim &P = a.prims[__float_as_uint(s8.z)];
    float bx = 1.0f - hu - hv, by = hu, bz = hv;
    V3 posv = (mk(s0.x, s0.y, s0.z) * bx + mk(s0.w, s1.x, s1.y) * by) + mk(s1.z, s1.w, s2.x) * bz;
    V3 world_pos = xform_point(P.o2w, posv);
    float tu = (s2.y * bx + s2.w * by) + s3.y * bz, tv = (s2.z * bx + s3.x * by) + s3.z * bz;
    V3 nrm = nrm3((mk(s3.w, s4.x, s4.y) * bx + mk(s4.z, s4.w, s5.x) * by) + mk(s5.y, s5.z, s5.w) * bz);
    const float *Wm = P.w2o;
    V3 world_normal = nrm3(mk(dot3(nrm, mk(Wm[0], Wm[4], Wm[8])), dot3(nrm, mk(Wm[1], Wm[5], Wm[9])), dot3(nrm, mk(Wm[2], Wm[6], Wm[10]))));
    V3 tan = nrm3((mk(s6.x, s6.y, s6.z) * bx + mk(s6.w, s7.x, s7.y) * by) + mk(s7.z, s7.w, s8.x) * bz);
    V3 world_tangent = nrm3(xform_vec(P.o2w, tan));
    world_tangent = nrm3(world_tangent - world_normal * dot3(world_tangent, world_normal));
    V3 world_binormal = cross3(world_normal, world_tangent) * s8.y;
    float4 tx = sample_tex(a.tex_pool, P, 2, tu, tv);
    V3 N = nrm3(mk(tx.x * 2.0f - 1.0f, tx.y * 2.0f - 1.0f, tx.z * 2.0f - 1.0f));
    N = nrm3((world_tangent * N.x + world_binormal * N.y) + world_normal * N.z);
    tx = sample_tex(a.tex_pool, P, 0, tu, tv);
    V3 albedo = mk(__powf(tx.x, 2.2f), __powf(tx.y, 2.2f), __powf(tx.z, 2.2f)); // radiance-only from here: fast intrinsics
    tx = sample_tex(a.tex_pool, P, 1, tu, tv);
    float roughness = tx.y, metallic = tx.z;
    S.world_pos = world_pos; S.N = N; S.albedo = albedo; S.metallic = metallic;
    S.Vv = nrm3(ld3(cam.camera_pos) - world_pos); // exact: V + L cancels at grazing angles and would amplify a 1-ulp rsq
    S.alpha = roughness * roughness;
    S.nc_NdotV = dot3(N, S.Vv);
    S.NdotV = clampf(S.nc_NdotV, 1e-5f, 1.0f);
    V3 vp = mat4_mul(cam.view, world_pos.x, world_pos.y, world_pos.z, 1.0f);
    out_depth = -vp.z;
    const float *VI = cam.view_inv;
    V3 on = mk((VI[0] * N.x + VI[1] * N.y) + VI[2] * N.z, (VI[4] * N.x + VI[5] * N.y) + VI[6] * N.z, (VI[8] * N.x + VI[9] * N.y) + VI[10] * N.z);
    on.y = -on.y; on.z = -on.z;
    on = nrm3(on);
    out_normal = mk(on.x * 0.5f + 0.5f, on.y * 0.5f + 0.5f, on.z * 0.5f + 0.5f);
}

// rgen:152-185 up to the shadow ray: c = (rho_s + rho_d) * radiance, c.w = NdotL; the shadow ray (origin, tmax | direction) if one is due
__device__ __forceinline__ bool shade_light(const ArtLight &l, const Surface &S, float4 &c4, float4 &ro, float4 &rd) {
    // a directional light's L and |nn_L| do not depend on the pixel: the host made them (art_api.hip directional_constants, the same operations)
    const bool directional = l.type == 2u;
    V3 nn_L = directional ? mk(0.f, 0.f, 0.f) : get_unnormalized_L_vec(l, S.world_pos);
    V3 L = directional ? ld3(l.area_pos2) : nrm3(nn_L);
    V3 Hh = nrm3(S.Vv + L);
    float nc_NdotL = dot3(S.N, L);
    float NdotL = clampf(nc_NdotL, 0.0f, 1.0f);
    float NdotH = clampf(dot3(S.N, Hh), 0.0f, 1.0f);
    float LdotH = clampf(dot3(L, Hh), 0.0f, 1.0f);
    float sch = pow5(1.0f - LdotH);
    V3 F0 = mk(mixf(0.04f, S.albedo.x, S.metallic), mixf(0.04f, S.albedo.y, S.metallic), mixf(0.04f, S.albedo.z, S.metallic));
    V3 Ks = mk(F0.x + (1.0f - F0.x) * sch, F0.y + (1.0f - F0.y) * sch, F0.z + (1.0f - F0.z) * sch);
    V3 Kd = S.albedo * (1.0f - S.metallic);
    float DG = D_GGX(S.alpha, NdotH) * V_SmithGGXCorrelated_fast(S.alpha, S.NdotV, NdotL);
    V3 rho_s = Ks * DG;
    V3 rho_d = Kd * Burley_diffuse_local_sss(S.alpha, S.NdotV, S.nc_NdotV, nc_NdotL, LdotH, 0.4f);
    V3 rad = get_light_radiance(l, S.world_pos, L);
    V3 c = (rho_s + rho_d) * rad;
    c4 = make_float4(c.x, c.y, c.z, NdotL);
    if (l.casts_shadows && nc_NdotL > 0.0f) { // raytrace.rgen.glsl:165: origin world_pos, dir L, tmax length(nn_L)
        ro = make_float4(S.world_pos.x, S.world_pos.y, S.world_pos.z, directional ? l.penumbra_angle : len3(nn_L));
        rd = make_float4(L.x, L.y, L.z, 0.f);
        return true;
    }
    return false;
}

// Light i of a frame: the first kMaxLights records travel by value in the kernel arguments (FrameArgs::lights), the rest -- the reference's list is a Vec, vk_lights.rs:89-91 --
// in a table of the frame's ring slot.  Both are read through the constant address space (wave-uniform, read-only: scalar loads, also behind the frame's own stores).
// (The argument copy is addressed through the kernel-argument segment pointer: FrameArgs is the first -- by-value -- argument of k_frame and k_shade, and taking the address of
// a.lights[i] itself would make the compiler copy all 2.6 KB of it to scratch.)
__device__ __forceinline__ ArtLight frame_light(const FrameArgs &a, uint32_t i) {
    ArtLight L;
#ifdef __HIP_DEVICE_COMPILE__
    typedef __attribute__((address_space(4))) const uint32_t *Words;
    const Words args = (Words)((__attribute__((address_space(4))) const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(FrameArgs, lights));
    const Words p = i < (uint32_t)kMaxLights ? args + i * (uint32_t)(sizeof(ArtLight) / 4) : (Words)(uintptr_t)(a.lights_more + (i - (uint32_t)kMaxLights));
    uint32_t w[sizeof(ArtLight) / 4];
#pragma unroll
    for (uint32_t k = 0; k < sizeof(ArtLight) / 4; k++) w[k] = p[k];
    __builtin_memcpy(&L, w, sizeof(ArtLight));
#else
    L = a.lights[i < (uint32_t)kMaxLights ? i : 0];   // (host pass of the compiler only)
#endif
    return L;
}
// staged frame, stage 2: emits one shadow ray per (pixel, light) that needs it
__global__ __launch_bounds__(kBlock) void k_shade(FrameArgs a) {
    if (blockIdx.x * kBlock >= a.n_local) return;
    uint32_t p = a.block_order[blockIdx.x] * kBlock + threadIdx.x;
    uint32_t x, y;
    bool in = local_to_xy(p, a.tile_list, a.tiles_x, a.W, a.H, x, y);
    size_t pix = (size_t)y * a.W + x;
    float4 h = ld_nt(&a.hits[p]);
    uint32_t pos = in ? __float_as_uint(h.w) : kNoHit;
    float out_depth = 10000.0f;
    V3 out_normal = mk(0.5f, 0.5f, 0.5f);
    uint32_t sbits = 0;
    if (pos == kNoHit) {
        for (uint32_t i = 0; i < a.n_lights; i++) {
            size_t slot = (size_t)i * a.n_local + p;
            st_nt(&a.contrib[slot], make_float4(0.f, 0.f, 0.f, 0.f));
            st_nt(&a.shadow_rays[2 * slot], make_float4(0.f, 0.f, 0.f, -1.0f)); // no shadow ray in this slot
        }
    } else {
        Surface S;
        shade_surface(a, a.cam, pos, h.y, h.z, S, out_depth, out_normal);
        for (uint32_t i = 0; i < a.n_lights; i++) {
            float4 c4, ro, rd;
            const ArtLight L = frame_light(a, i);
            bool want = shade_light(L, S, c4, ro, rd);
            size_t slot = (size_t)i * a.n_local + p;
            st_nt(&a.contrib[slot], c4);
            if (want) {
                st_nt(&a.shadow_rays[2 * slot], ro);
                st_nt(&a.shadow_rays[2 * slot + 1], rd);
                if (i < 16) sbits |= 1u << (16 + i);
            } else st_nt(&a.shadow_rays[2 * slot], make_float4(0.f, 0.f, 0.f, -1.0f));
        }
    }
    if (in) {
        st_nt(&a.depth[pix], out_depth);
        st_nt(&a.normal[pix], make_float4(out_normal.x, out_normal.y, out_normal.z, 1.0f));
    }
    if (a.shadow_bits) a.shadow_bits[p] = sbits;
    uint64_t hitmask = __ballot(pos != kNoHit); // hit-pixel count: one atomic per wave, off the critical path
    if ((threadIdx.x & 63u) == 0 && hitmask) atomicAdd(&a.counters[kHitSlots + ((blockIdx.x * 4u + (threadIdx.x >> 6)) % kSlotCount) * kSlotStride], (uint32_t)__popcll(hitmask));
}

// The fused frame: one wave takes an 8x8 pixel block through the whole of raytrace.rgen.glsl -- primary packet, shading, one
// shadow packet per light, accumulation -- and writes only the frame's outputs.  No hit / shadow-ray / contribution records
// cross HBM, one launch per frame, and the CUs hold waves in every phase at once (node-fetch latency of the walks, texture
// latency and ALU of the shading), which is what the staged frame needed a dozen frames in flight for.  The arithmetic and its
// order are the staged kernels': the frames are bit-identical.  pix_bits[p]: bit i = light i shadowed, bit 16+i = shadow ray
// traced (art_get_stats counts rays from it on demand; art_read_shadow_bits).
// the same value, but the compiler cannot know it: what is computed from it is computed again instead of being kept in registers.
// (A "memory" clobber would do that too, and would turn every wave-uniform node fetch after it from a scalar into a vector load.)
__device__ __forceinline__ uint32_t launder(uint32_t v) { asm volatile("" : "+s"(v)); return v; }   // wave-uniform values
__device__ __forceinline__ uint32_t launder_v(uint32_t v) { asm volatile("" : "+v"(v)); return v; } // per-lane values
// what lane `__lane_id()` of wave item `wid` traces: local pixel id, frame coordinates, whether the item's cell mask covers it; returns "traces a ray"
__device__ __forceinline__ bool frame_pixel(const FrameArgs &a, uint32_t wid, uint32_t &p, uint32_t &x, uint32_t &y, bool &mine) {
    // wave-uniform, read-only tables: constant address space, so these stay scalar loads behind the frame's stores too (no divisions either)
    typedef uint32_t U2 __attribute__((ext_vector_type(2)));
    const U2 item_ = ((__attribute__((address_space(4))) const U2 *)(uintptr_t)a.wave_items)[wid];
    const uint2 item = make_uint2(item_.x, item_.y);
    const uint32_t txy = ((__attribute__((address_space(4))) const uint32_t *)(uintptr_t)a.tile_xy)[item.x >> 4];    // the block's 32x32 tile: x | y << 16
    const uint32_t lane = __lane_id(), sub = item.x & 15u;
    p = item.x * 64u + lane;
    mine = (item.y >> (((lane >> 4) << 2) | ((lane >> 1) & 3u))) & 1u; // cell = (y/2)*4 + x/2 of the 8x8 block
    x = (txy & 0xFFFFu) * kTile + (sub & 3u) * 8u + (lane & 7u);
    y = (txy >> 16) * kTile + (sub >> 2) * 8u + (lane >> 3);
    return x < a.W && y < a.H && mine;
}
#ifdef ART_PHASE_PROF
// profiling build only (make EXTRA=-DART_PHASE_PROF; tools/phase_prof.py): shader-clock cycles a wave spends in each phase of k_frame, summed over waves
constexpr uint32_t kPhaseWaves = 1u << 18;
__device__ uint32_t g_phase[8][kPhaseWaves]; // [phase][wave item]: the last frame that ran wrote it
__device__ __forceinline__ uint64_t tick(float &dep) { uint64_t t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep)); return t; }
#define PHASE(i, dep) { uint64_t t1_ = tick(dep); if (__lane_id() == 0 && blockIdx.x < kPhaseWaves) g_phase[i][blockIdx.x] = (uint32_t)(t1_ - t0_); t0_ = t1_; }
#else
#define PHASE(i, dep)
#endif
template <bool WIDE, bool ONE_LIGHT, bool COUNT = false, bool BATCH = false>   // WIDE: the 128-byte 4-wide nodes (the default) | the 64-byte binary nodes
__global__ __launch_bounds__(kFrameBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_frame(FrameArgs a) {
    // One wave per workgroup: the waves of a frame are independent (nothing is shared, no barrier), and a workgroup of four held its LDS and its place
    // in the dispatcher's books until its slowest wave was done -- packets differ 25x in steps.  Single-wave groups: +2.5 % rays/s (profiles/README.md r2).
    __shared__ int wstack[kPacketStack];
    int *stk = wstack;
    const uint32_t wid = blockIdx.x;
    if (wid >= a.n_wave_items) return;
    uint32_t steps = 0; // packet steps of this wave, all walks
    const uint32_t fb = BATCH ? blockIdx.y : 0u;   // which frame of the launch (wave-uniform)
    const CameraArg &cam = (BATCH && fb) ? a.cam_more[fb - 1] : a.cam;
    const size_t frame_px = (size_t)fb * a.W * a.H, frame_local = (size_t)fb * a.n_local;
    // The pixel a lane works on is looked up again wherever it is needed (here, at the depth/normal stores, at the end) instead of being
    // carried through the walks: the walks run at the 64-register edge.
    bool in;
    Ray r;
#ifdef ART_PHASE_PROF
    float dep0_ = 0.f; uint64_t t0_ = tick(dep0_);
#endif
    {
        uint32_t p, x, y; bool mine;
        in = frame_pixel(a, wid, p, x, y, mine);
        float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
        float dx = (fx / (float)a.W) * 2.0f - 1.0f, dy = (fy / (float)a.H) * 2.0f - 1.0f;
        V3 org = mat4_mul(cam.view_inv, 0.f, 0.f, 0.f, 1.f);
        V3 tgt = nrm3(mat4_mul(cam.proj_inv, dx, dy, 1.f, 1.f));
        V3 dir = mat4_mul(cam.view_inv, tgt.x, tgt.y, tgt.z, 0.f);
        ray_init(r, org, dir, 0.001f, 10000.0f);
    }
    bool on = in && ray_finite(r.o, r.d);
    PHASE(0, r.inv.x)
    float tbest = on ? r.tmax : -1.0f, bu = 0.f, bv = 0.f;
    uint32_t bpos = kNoHit, bgid = kNoHit;
    walk_dispatch<false, WIDE, COUNT>(a, r, on, stk, tbest, bu, bv, bpos, bgid, steps);
    PHASE(1, tbest)
    uint32_t p, x, y; bool mine;
    frame_pixel(a, wid, p, x, y, mine);
    if (a.keep_hits && mine) a.hits[frame_local + p] = bpos != kNoHit ? make_float4(tbest, bu, bv, __uint_as_float(bpos)) : make_float4(10000.0f, 0.f, 0.f, __uint_as_float(kNoHit));
    const bool hit = in && bpos != kNoHit;
    float out_depth = 10000.0f;
    V3 out_normal = mk(0.5f, 0.5f, 0.5f);
    Surface S;
    S.world_pos = mk(0.f, 0.f, 0.f); S.N = mk(0.f, 0.f, 1.f); S.Vv = mk(0.f, 0.f, 1.f); S.albedo = mk(0.f, 0.f, 0.f);
    S.metallic = 0.f; S.alpha = 0.f; S.nc_NdotV = 0.f; S.NdotV = 0.f;
    if (hit) shade_surface(a, cam, bpos, bu, bv, S, out_depth, out_normal);
    if (in) {
        const size_t pix = frame_px + (size_t)y * a.W + x;
        st_nt(&a.depth[pix], out_depth);
        st_nt(&a.normal[pix], make_float4(out_normal.x, out_normal.y, out_normal.z, 1.0f));
    }
    PHASE(2, S.NdotV)
    float rx = 0.f, ry = 0.f, rz = 0.f;
    uint32_t sbits = 0, more = 0;
    // (until the walks fetched their nodes into SGPRs the multi-light instance parked the surface record in LDS across each shadow walk; it fits now)
    for (uint32_t i = 0; i < (ONE_LIGHT ? 1u : a.n_lights); i++) { // uniform loop: the shadow packet needs the whole wave
        float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f), ro = make_float4(0.f, 0.f, 0.f, 1.0f), rd = make_float4(0.f, 0.f, 1.f, 0.f);
        bool want = false;
        const ArtLight L = ONE_LIGHT ? a.lights[0] : frame_light(a, i);
        if (hit) want = shade_light(L, S, c4, ro, rd);
        if (want) { if (i < 16u) sbits |= 1u << (16u + i); else more++; }   // (lights 16.. have no bits of their own: their shadow rays are counted)
        Ray sr;
        if (L.type == 2u) ray_init_inv(sr, mk(ro.x, ro.y, ro.z), ld3(L.area_pos2), ld3(L.area_pos3), 0.01f, L.penumbra_angle); // (wave-uniform branch)
        else ray_init(sr, mk(ro.x, ro.y, ro.z), mk(rd.x, rd.y, rd.z), 0.01f, ro.w);
        bool son = want && ray_finite(sr.o, sr.d);
        PHASE(3, sr.inv.x)
        float st = son ? sr.tmax : -1.0f, su = 0.f, sv = 0.f;
        uint32_t spos = kNoHit, sgid = kNoHit;
        walk_dispatch<true, WIDE, COUNT>(a, sr, son, stk, st, su, sv, spos, sgid, steps);
        PHASE(4, st)
        if (want && spos != kNoHit) { // shadowed: the light keeps 0.05 of its contribution (raytrace.rgen.glsl:179-181)
            c4 = make_float4(c4.x * 0.05f, c4.y * 0.05f, c4.z * 0.05f, c4.w);
            if (i < 16u) sbits |= 1u << i;
        }
        rx += c4.x * c4.w; ry += c4.y * c4.w; rz += c4.z * c4.w; // rgen:185, lights in order
    }
    if (!in) { rx = 0.f; ry = 0.f; rz = 0.f; }
    float4 o = make_float4(rx, ry, rz, 1.0f);
    frame_pixel(a, wid, p, x, y, mine);
    if (in) st_nt(&a.color[frame_px + (size_t)y * a.W + x], o);
    if (a.color_tiles && mine) { // compact tile buffer for the gather: row-major inside each 32x32 tile
        uint32_t q = p & 1023u, sub = q >> 6, l = q & 63u;
        uint32_t lx = (sub & 3u) * 8u + (l & 7u), ly = (sub >> 2) * 8u + (l >> 3);
        size_t ti = (size_t)fb * a.tiles_stride + (size_t)(p >> 10) * kTilePixels + ly * kTile + lx;
        if (a.tiles_packed) __builtin_nontemporal_store(pack_b10g11r11(o.x, o.y, o.z), reinterpret_cast<uint32_t *>(a.color_tiles) + ti);
        else st_nt_rgb(a.color_tiles, ti, o);
    }
    if (mine) a.pix_bits[frame_local + p] = sbits;
    if (!ONE_LIGHT && a.pix_more && mine) a.pix_more[frame_local + p] = more;   // more than 16 lights: shadow rays of lights 16.. (art_get_stats)
    PHASE(5, rx)
#ifdef ART_PHASE_PROF
    if (__lane_id() == 0 && blockIdx.x < kPhaseWaves) { g_phase[6][blockIdx.x] = 1u; g_phase[7][blockIdx.x] = steps; }
#endif
    if (COUNT && __lane_id() == 0) a.wave_cost[wid] = steps; // feedback for the next plan (art_api.hip plan_poll)
}

// ---- the wave plan, on the device ---------------------------------------------------------------------------------------------------------------------------------
// Rounds 1-3 made it on the host: the step counts of a sampled frame travelled up, one thread went through 32 640 blocks and built the next table (0.3-0.9 ms inside an
// art_trace call every few frames once the camera moves: frames are 0.16 ms), the table travelled down.  One workgroup does the same in ~20 us behind the sampled frame on
// that frame's stream; the host only learns "a new table of n items is ready" from pinned memory once the event behind this launch has fired.
// Table layout (what plan_build_items made): the split blocks' parts first -- sixteen cells each, then four quadrants each: the long poles start first -- padded to whole groups
// of four, then every 256-pixel block of the launch order as four 8x8 items (mask 0: an idle wave where the block is split); finally the deal that keeps the four waves of
// launch block 8g + x on XCD x (workgroup j runs on XCD j % 8): position 32g + 4x + k goes to 32g + 8k + x.
__device__ __forceinline__ uint32_t plan_block_sum(uint32_t v, uint32_t *sh) {   // sum over the 1024 threads, in every thread (sh: 16 words)
    for (int off = 32; off >= 1; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t t = 0;
    for (int i = 0; i < 16; i++) t += sh[i];
    return t;
}
__device__ __forceinline__ uint32_t plan_deal(uint32_t j, uint32_t total) { const uint32_t g = j >> 5; return (g + 1u) * 32u <= total ? g * 32u + 8u * (j & 3u) + ((j & 31u) >> 2) : j; }
__global__ __launch_bounds__(1024) void k_plan(PlanArgs p) {
    __shared__ uint32_t sh[16], sc2[1024], sc1[1024];
    __shared__ unsigned long long sh64[16];
    const uint32_t tid = threadIdx.x;
    // 1. the slowest wave of every block in the sampled frame, and the frame's steps
    unsigned long long sum = 0;
    for (uint32_t i = tid; i < p.n_items_in; i += 1024u) { const uint2 it = p.items_in[i]; if (it.y) { const uint32_t c = p.cost[i]; atomicMax(&p.worst[it.x], c); sum += c; } }
    for (int off = 32; off >= 1; off >>= 1) sum += (unsigned long long)__shfl_xor((long long)sum, off);
    if ((tid & 63u) == 0) sh64[tid >> 6] = sum;
    __threadfence_block();
    __syncthreads();
    sum = 0;
    for (int i = 0; i < 16; i++) sum += sh64[i];
    uint32_t T = p.fixed_steps ? p.fixed_steps : max(p.min_steps, (uint32_t)(p.share * (float)sum));
    // 2. every block's level; a block that has to be split is a wave that outlasts its launch, blocks that could be put together again only save a few steps (going down needs
    //    a clear margin: the parts of a split block share the top of their walks)
    const uint32_t per = (p.n64 + 1023u) / 1024u, b0 = min(tid * per, p.n64), b1 = min(b0 + per, p.n64);
    uint32_t n1 = 0, n2 = 0, N1 = 0, N2 = 0, ups = 0, downs = 0, split = 0, wmax = 0;
    for (int attempt = 0; attempt < 8; attempt++, T += T / 2) {
        n1 = n2 = 0; uint32_t u = 0, d = 0, sp = 0;
        for (uint32_t b = b0; b < b1; b++) {
            const uint32_t w = __hip_atomic_load(&p.worst[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (written by atomics of other waves of this workgroup)
            const uint32_t old = p.level[b];
            uint32_t lv = old;
            if (old == 0) lv = w > 4u * T ? 2u : (w > T ? 1u : 0u);
            else if (old == 1) lv = w > T ? 2u : (w * 4u < T / 2u ? 0u : 1u);
            else lv = w * 4u < T / 2u ? 1u : 2u;
            p.level_tmp[b] = (uint8_t)lv;
            n1 += lv == 1u; n2 += lv == 2u; u += lv > old; d += lv < old; sp += old != 0u; wmax = max(wmax, w);
        }
        N1 = plan_block_sum(n1, sh); N2 = plan_block_sum(n2, sh); ups = plan_block_sum(u, sh); downs = plan_block_sum(d, sh); split = plan_block_sum(sp, sh);
        if (p.n64 + (p.n64 & 3u) + 4u * N1 + 16u * N2 + 4u <= p.cap) break;
        if (attempt == 7) { N1 = N2 = ups = downs = 0; for (uint32_t b = b0; b < b1; b++) p.level_tmp[b] = p.level[b]; n1 = n2 = 0; }   // (never seen: no target fits -- the plan stays)
    }
    for (int off = 32; off >= 1; off >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, off));
    __syncthreads();
    if ((tid & 63u) == 0) sh[tid >> 6] = wmax;
    __syncthreads();
    for (int i = 0; i < 16; i++) wmax = max(wmax, sh[i]);
    for (uint32_t b = b0; b < b1; b++) p.worst[b] = 0u;   // clear for the next sample
    const bool changed = p.fixed_steps ? (ups + downs) != 0u : (ups != 0u || downs > max(16u, split / 4u));
    if (!changed) { if (tid == 0) { p.result[1] = 0u; p.result[4] = T; p.result[5] = wmax; } return; }
    // 3. where every thread's split blocks go: exclusive scans of the per-thread counts
    sc2[tid] = n2; sc1[tid] = n1;
    __syncthreads();
    uint32_t base2 = 0, base1 = 0;
    for (uint32_t i = 0; i < tid; i++) { base2 += sc2[i]; base1 += sc1[i]; }   // (1024 LDS reads a thread: a microsecond)
    const uint32_t heavy = 16u * N2 + 4u * N1, heavy_pad = (heavy + 3u) & ~3u, total = heavy_pad + p.n256 * 4u;
    for (uint32_t b = b0; b < b1; b++) {
        const uint32_t lv = p.level_tmp[b];
        p.level[b] = (uint8_t)lv;
        if (lv == 2u) { for (uint32_t cell = 0; cell < 16u; cell++) p.items_out[plan_deal(16u * base2 + cell, total)] = make_uint2(b, 1u << cell); base2++; }
        else if (lv == 1u) { const uint32_t quad[4] = {0x0033u, 0x00CCu, 0x3300u, 0xCC00u}; for (uint32_t q = 0; q < 4u; q++) p.items_out[plan_deal(16u * N2 + 4u * base1 + q, total)] = make_uint2(b, quad[q]); base1++; }
    }
    for (uint32_t j = heavy + tid; j < heavy_pad; j += 1024u) p.items_out[plan_deal(j, total)] = make_uint2(0u, 0u);
    __threadfence_block();
    __syncthreads();   // (level[] of other threads' blocks is read below)
    // 4. the blocks in launch order; a split block leaves an idle wave behind: the XCD order of the rest is untouched
    for (uint32_t r = tid; r < p.n256 * 4u; r += 1024u) {
        const uint32_t b = p.order[r >> 2] * 4u + (r & 3u);
        p.items_out[plan_deal(heavy_pad + r, total)] = make_uint2(b, p.level[b] ? 0u : 0xFFFFu);
    }
    if (tid == 0) { p.result[0] = total; p.result[1] = 1u; p.result[2] = N1; p.result[3] = N2; p.result[4] = T; p.result[5] = wmax; }
}
void launch_plan(const PlanArgs &p, hipStream_t s) { k_plan<<<1, 1024, 0, s>>>(p); }

// art_get_stats for fused frames: shadow rays = set bits 16..31 of pix_bits, hit pixels = depth < miss depth; on demand only
__global__ __launch_bounds__(kBlock) void k_frame_stats(FrameArgs a, uint32_t *out) {
    uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    uint32_t rays = 0, hits = 0;
    if (p < a.n_local) {
        uint32_t x, y;
        bool in = local_to_xy(p, a.tile_list, a.tiles_x, a.W, a.H, x, y);
        rays = (uint32_t)__popc(a.pix_bits[p] >> 16) + (a.pix_more ? a.pix_more[p] : 0u);
        hits = in && a.depth[(size_t)y * a.W + x] < 10000.0f ? 1u : 0u;
    }
    for (int off = 32; off >= 1; off >>= 1) { rays += (uint32_t)__shfl_xor((int)rays, off); hits += (uint32_t)__shfl_xor((int)hits, off); }
    if ((threadIdx.x & 63u) == 0) { if (rays) atomicAdd(&out[0], rays); if (hits) atomicAdd(&out[1], hits); }
}

// rho += (rho_s + rho_d) * radiance * shadow_attenuation * NdotL (raytrace.rgen.glsl:185), lights in order
__global__ __launch_bounds__(kBlock) void k_accumulate(FrameArgs a) {
    if (blockIdx.x * kBlock >= a.n_local) return;
    uint32_t p = a.block_order[blockIdx.x] * kBlock + threadIdx.x;
    uint32_t x, y;
    bool in = local_to_xy(p, a.tile_list, a.tiles_x, a.W, a.H, x, y);
    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (in)
        for (uint32_t i = 0; i < a.n_lights; i++) {
            float4 c = ld_nt(&a.contrib[(size_t)i * a.n_local + p]);
            rx += c.x * c.w; ry += c.y * c.w; rz += c.z * c.w;
        }
    float4 o = make_float4(rx, ry, rz, 1.0f);
    if (in) st_nt(&a.color[(size_t)y * a.W + x], o);
    if (a.color_tiles) { // compact tile buffer for the gather: row-major inside each 32x32 tile
        uint32_t q = p & 1023u, sub = q >> 6, l = q & 63u;
        uint32_t lx = (sub & 3u) * 8u + (l & 7u), ly = (sub >> 2) * 8u + (l >> 3);
        size_t ti = (size_t)(p >> 10) * kTilePixels + ly * kTile + lx;
        if (a.tiles_packed) __builtin_nontemporal_store(pack_b10g11r11(o.x, o.y, o.z), reinterpret_cast<uint32_t *>(a.color_tiles) + ti);
        else st_nt_rgb(a.color_tiles, ti, o);
    }
}

// root side of the gather: shard s's j-th tile sits at gathered[(s*padded + j) * 1024]
// n_frames frames per launch (grid.z): frame z reads its shards' tiles frame_stride tiles further on and writes the z-th output
// frame.  A thread moves V consecutive pixels of one tile row (16 bytes when the width allows): no divisions, one table lookup.
template <class T, int V> __global__ __launch_bounds__(kBlock) void k_untile(const T *__restrict__ gathered, const uint32_t *__restrict__ tile_slot, uint32_t shard_stride, uint32_t frame_stride,
                                                                             uint32_t W, uint32_t H, T *__restrict__ frame) {
    // A block moves a band of 32 x V pixels by 32 rows = one row of tiles: a thread takes V pixels of four rows of ONE tile (one table
    // lookup, four independent loads, then four stores).  A quarter of the waves of a row per thread: on a root that keeps tracing,
    // the un-tile of 20 frames was 162 000 one-kilobyte waves competing with the frames' 82 000 for slots.
    const uint32_t tiles_x = (W + kTile - 1) / kTile;
    const uint32_t x = (blockIdx.x * 32u + (threadIdx.x & 31u)) * V, y0 = blockIdx.y * kTile + (threadIdx.x >> 5);
    if (x >= W || y0 >= H) return;
    const uint32_t ts = tile_slot[blockIdx.y * tiles_x + x / kTile]; // owner << 24 | index among the owner's tiles (host table, setup_frame)
    const size_t tile = (size_t)(ts >> 24) * shard_stride + (size_t)blockIdx.z * frame_stride + (ts & 0xFFFFFFu); // shard_stride: tiles between two shards' buffers
    const T *src = gathered + tile * kTilePixels + (threadIdx.x >> 5) * kTile + (x % kTile);
    T *dst = frame + (size_t)blockIdx.z * W * H + (size_t)y0 * W + x;
    struct alignas(sizeof(T) * V) Pack { T v[V]; };
    Pack p[4];
#pragma unroll
    for (int i = 0; i < 4; i++) p[i] = *reinterpret_cast<const Pack *>(src + (size_t)i * 8u * kTile); // rows y0, y0 + 8, + 16, + 24 of the tile (padding rows of a bottom tile are readable)
#pragma unroll
    for (int i = 0; i < 4; i++) if (y0 + 8u * i < H) *reinterpret_cast<Pack *>(dst + (size_t)i * 8u * W) = p[i];
}

// the same for RGB32F tiles (12 B per texel) into the RGBA32F frame: alpha = 1, the constant the colour output carries
__global__ __launch_bounds__(kBlock) void k_untile_rgb(const float *__restrict__ gathered, const uint32_t *__restrict__ tile_slot, uint32_t shard_stride, uint32_t frame_stride,
                                                       uint32_t W, uint32_t H, float4 *__restrict__ frame) {
    const uint32_t tiles_x = (W + kTile - 1) / kTile;
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u), y0 = blockIdx.y * kTile + (threadIdx.x >> 5);
    if (x >= W || y0 >= H) return;
    const uint32_t ts = tile_slot[blockIdx.y * tiles_x + x / kTile];
    const size_t tile = (size_t)(ts >> 24) * shard_stride + (size_t)blockIdx.z * frame_stride + (ts & 0xFFFFFFu);
    const float *src = gathered + 3 * (tile * kTilePixels + (threadIdx.x >> 5) * kTile + (x % kTile));
    float4 *dst = frame + (size_t)blockIdx.z * W * H + (size_t)y0 * W + x;
    float4 p[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { const float *q = src + (size_t)i * 8u * kTile * 3; p[i] = make_float4(q[0], q[1], q[2], 1.0f); }   // rows y0, y0 + 8, + 16, + 24 of the tile
#pragma unroll
    for (int i = 0; i < 4; i++) if (y0 + 8u * i < H) dst[(size_t)i * 8u * W] = p[i];
}

// ------------------------------------------------------------------------------------------------ launchers
static inline uint32_t blocks_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }
// persistent grid: enough waves to fill the chip (8 blocks of 4 waves per CU), never more than the work needs
static inline uint32_t persistent_blocks(uint32_t total, const Tune &t) {
    uint32_t need = (total + kTraceBlock - 1) / kTraceBlock, cap = t.blocks * (kBlock / kTraceBlock);   // the presets count 256-thread blocks
    return need < cap ? (need ? need : 1u) : cap;
}
template <int MODE> static void launch_trace(TraceArgs &a, int kind, bool pipelined, const TraceTune &o, hipStream_t s) {
    const Tune t = tune(pipelined, MODE == MODE_AO, o);
    uint32_t nb = persistent_blocks(a.total, t);
    a.chunk = t.chunk; a.refill = t.refill; a.leaf_batch = t.leaf_batch;
    if (kind == 4) k_trace<MODE, 4><<<nb, kTraceBlock, 0, s>>>(a);
    else k_trace<MODE, 2><<<nb, kTraceBlock, 0, s>>>(a);
}
void launch_primary(const FrameArgs &f, hipStream_t s) {   // staged frames: the persistent per-ray tracer
    TraceArgs a{};
    a.nodes = f.nodes; a.wide = f.wide; a.tris = f.tris; a.total = f.n_local; a.cursors = f.counters + 64; a.cam = f.cam; a.W = f.W; a.H = f.H;
    a.tile_list = f.tile_list; a.tiles_x = f.tiles_x; a.hits = f.hits;
    launch_trace<MODE_PRIMARY>(a, f.trace_kind[0], f.pipelined, f.tune, s);
}
void launch_shade(const FrameArgs &a, hipStream_t s) { k_shade<<<blocks_for(a.n_local), kBlock, 0, s>>>(a); }
void launch_shadow(const FrameArgs &f, hipStream_t s) {
    if (f.n_lights == 0) return;
    TraceArgs a{};
    a.nodes = f.nodes; a.wide = f.wide; a.tris = f.tris; a.total = f.n_local * f.n_lights; a.cursors = f.counters + 64 + 8 * kCursorStride; a.count = f.counters + kShadowSlots;
    a.rays = f.shadow_rays; a.contrib = f.contrib; a.n_local = f.n_local; a.shadow_bits = f.shadow_bits;
    launch_trace<MODE_SHADOW>(a, f.trace_kind[1], f.pipelined, f.tune, s);
}
template <bool WIDE, bool ONE_LIGHT> static void launch_frame_form(const FrameArgs &a, uint32_t g, bool count, hipStream_t s) {
    if (a.batch > 1) { // several frames per launch
        const dim3 gb(g, a.batch);
        if (count) k_frame<WIDE, ONE_LIGHT, true, true><<<gb, kFrameBlock, 0, s>>>(a); else k_frame<WIDE, ONE_LIGHT, false, true><<<gb, kFrameBlock, 0, s>>>(a);
    } else if (count) k_frame<WIDE, ONE_LIGHT, true><<<g, kFrameBlock, 0, s>>>(a);
    else k_frame<WIDE, ONE_LIGHT><<<g, kFrameBlock, 0, s>>>(a);
}
bool launch_frame(const FrameArgs &a, hipStream_t s) { // returns whether the launch wrote a.wave_cost
    const uint32_t g = a.n_wave_items;   // one workgroup per wave item
    if (g == 0) return false;
    const bool one = a.n_lights == 1;
    const bool count = a.wave_cost != nullptr && a.n_lights > 0; // a sampled frame of the wave plan: the step-counting instances
    if (a.packet_wide) { if (one) launch_frame_form<true, true>(a, g, count, s); else launch_frame_form<true, false>(a, g, count, s); }
    else { if (one) launch_frame_form<false, true>(a, g, count, s); else launch_frame_form<false, false>(a, g, count, s); }
    return count;
}
void launch_frame_stats(const FrameArgs &a, uint32_t *out, hipStream_t s) { k_frame_stats<<<blocks_for(a.n_local), kBlock, 0, s>>>(a, out); }
void launch_accumulate(const FrameArgs &a, hipStream_t s) { k_accumulate<<<blocks_for(a.n_local), kBlock, 0, s>>>(a); }
// queries: rays[2i] = o.xyz,tmin | rays[2i+1] = d.xyz,tmax;  cursors: 8 * kCursorStride zeroed words
void launch_query_closest(const BvhView &b, const float4 *rays, uint32_t n, float4 *hits, uint32_t *cursors, hipStream_t s) {
    if (!n) return;
    TraceArgs a{};
    a.nodes = b.nodes; a.wide = b.wide; a.tris = b.tris; a.total = n; a.cursors = cursors; a.rays = rays; a.hits = hits;
    launch_trace<MODE_QUERY_CLOSEST>(a, b.kind, false, b.tune, s);
}
void launch_query_any(const BvhView &b, const float4 *rays, uint32_t n, uint32_t *hit, uint32_t *cursors, hipStream_t s) {
    if (!n) return;
    TraceArgs a{};
    a.nodes = b.nodes; a.wide = b.wide; a.tris = b.tris; a.total = n; a.cursors = cursors; a.rays = rays; a.any_out = hit;
    launch_trace<MODE_QUERY_ANY>(a, b.kind, false, b.tune, s);
}
// AO resolve: occluded count -> uint(pow(visibility, 2.2) * 255 + 0.5) through a host-built table; 255 where nothing was hit
#ifdef ART_PACKET_PROF
extern "C" int32_t art_debug_packet_prof(unsigned long long *out, int32_t reset) { // out[24]
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_packet_prof), sizeof(g_packet_prof)) != hipSuccess) return -1;
    void *dp = nullptr;
    if (reset && (hipGetSymbolAddress(&dp, HIP_SYMBOL(g_packet_prof)) != hipSuccess || hipMemset(dp, 0, sizeof(g_packet_prof)) != hipSuccess)) return -1;
    return 0;
}
#endif
#ifdef ART_TRACE_PROF
extern "C" int32_t art_debug_trace_prof(unsigned long long *out, int32_t reset) { // out[8]
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace_prof), sizeof(g_trace_prof)) != hipSuccess) return -1;
    void *dp = nullptr;
    if (reset && (hipGetSymbolAddress(&dp, HIP_SYMBOL(g_trace_prof)) != hipSuccess || hipMemset(dp, 0, sizeof(g_trace_prof)) != hipSuccess)) return -1;
    return 0;
}
#endif
#ifdef ART_PHASE_PROF
extern "C" int32_t art_debug_phase(uint32_t *out, int32_t reset) { // out[8][2^18]
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(g_phase)) != hipSuccess) return -1;
    void *dp = nullptr;
    if (reset && (hipGetSymbolAddress(&dp, HIP_SYMBOL(g_phase)) != hipSuccess || hipMemset(dp, 0, sizeof(g_phase)) != hipSuccess)) return -1;
    return 0;
}
#endif
struct AoLut { uint32_t v[65]; };
__global__ __launch_bounds__(kBlock) void k_ao_resolve(FrameArgs a, const uint8_t *__restrict__ occl, uint32_t spp, AoLut lut, uint32_t *__restrict__ ao) { // occl: one byte per slot (ao_slot_of)
    uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= a.n_local) return;
    uint32_t x, y;
    if (!local_to_xy(p, a.tile_list, a.tiles_x, a.W, a.H, x, y)) return;
    size_t pix = (size_t)y * a.W + x;
    uint32_t k = 0;
    for (uint32_t s = 0; s < spp; s++) k += occl[ao_slot_of(p, s, spp)];
    ao[pix] = a.depth[pix] < 10000.0f ? lut.v[k] : 255u;
}
void launch_ao_table(uint32_t spp, float4 *tab, hipStream_t s) { k_ao_table<<<blocks_for(kAoNoiseTile), kBlock, 0, s>>>(spp, tab); }
void launch_ao(const FrameArgs &f, uint32_t spp, float radius, uint8_t *occl, float4 *pix, const float4 *tab, bool entry_search, uint32_t *ao, const uint32_t *lut, hipStream_t s) {
    AoLut l; for (uint32_t k = 0; k < 65; k++) l.v[k] = lut[k];
    k_ao_pixels<<<blocks_for(f.n_local), kBlock, 0, s>>>(f, ((f.trace_kind[2] == 4 || f.trace_kind[2] == 6) && entry_search) ? f.wide : nullptr, radius, pix); // one point, normal and entry node per pixel for its spp rays
    const uint32_t n_slots = (f.n_local / 16u) * (((spp + 3u) >> 2) * 64u); // ao_slot_decode
    TraceArgs a{};
    a.nodes = f.nodes; a.wide = f.wide; a.tris = f.tris; a.total = n_slots; a.cursors = f.counters + 64 + 16 * kCursorStride; a.cam = f.cam; a.W = f.W; a.H = f.H;
    a.tile_list = f.tile_list; a.tiles_x = f.tiles_x; a.depth = f.depth; a.normal = f.normal; a.spp = spp; a.ao_radius = radius; a.occl = occl;
    a.ao_pix = pix; a.ao_tab = tab;
    if (f.trace_kind[2] == 4) {   // the default: the AO launch's own tracer (rays made by the whole wave into a pool); 6 = the same walk through the generic tracer (round 3's form), 2 = binary nodes
        Tune t = tune(f.pipelined, true, f.tune);
        if (!(f.tune.refill >= 1 && f.tune.refill <= 64)) t.refill = kAoPoolTake;
        a.chunk = t.chunk; a.refill = t.refill; a.leaf_batch = t.leaf_batch;
        k_trace_ao<<<persistent_blocks(a.total, t), kTraceBlock, 0, s>>>(a);
    } else launch_trace<MODE_AO>(a, f.trace_kind[2] == 6 ? 4 : f.trace_kind[2], f.pipelined, f.tune, s);
    k_ao_resolve<<<blocks_for(f.n_local), kBlock, 0, s>>>(f, occl, spp, l, ao);
}
void launch_untile(const float4 *gathered, const uint32_t *tile_slot, uint32_t shard_stride, uint32_t n_frames, uint32_t frame_stride, uint32_t W, uint32_t H, float4 *frame, hipStream_t s) {
    dim3 g((W + 31) / 32, (H + kTile - 1) / kTile, n_frames);
    k_untile_rgb<<<g, kBlock, 0, s>>>(reinterpret_cast<const float *>(gathered), tile_slot, shard_stride, frame_stride, W, H, frame);
}
void launch_untile_packed(const uint32_t *gathered, const uint32_t *tile_slot, uint32_t shard_stride, uint32_t n_frames, uint32_t frame_stride, uint32_t W, uint32_t H, uint32_t *frame, hipStream_t s) {
    if (W % 4 == 0 && ((uintptr_t)frame & 15u) == 0 && ((uintptr_t)gathered & 15u) == 0) { // four pixels (16 bytes) per thread
        dim3 g((W / 4 + 31) / 32, (H + kTile - 1) / kTile, n_frames);
        k_untile<uint32_t, 4><<<g, kBlock, 0, s>>>(gathered, tile_slot, shard_stride, frame_stride, W, H, frame);
    } else {
        dim3 g((W + 31) / 32, (H + kTile - 1) / kTile, n_frames);
        k_untile<uint32_t, 1><<<g, kBlock, 0, s>>>(gathered, tile_slot, shard_stride, frame_stride, W, H, frame);
    }
}

} // namespace art
