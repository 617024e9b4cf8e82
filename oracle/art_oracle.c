/*
 * art_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See art_oracle.h.
 * PARITY UNPINNED (no reference-authored golden vectors exist for this path).
 *
 * Build: gcc -O2 -fno-tree-vectorize -ffp-contract=off -mfma -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off + explicit fmaf() fixes the floating-point expression order, so that the
 * geometry stages (ray generation, slab test, Moller-Trumbore, hit reconstruction up to the shadow
 * ray) are reproducible bit for bit by any implementation that follows the same expressions.
 *
 * Geometry semantics defined here (the Vulkan driver's are closed; SURVEY.md appendix A "define"):
 *   accept(tri, ray)  :=  slab(AABB(tri), ray) passes   AND   Moller-Trumbore hits with tmin < t < tmax
 *   t_eff             :=  max(t_MT, t_entry(AABB(tri)))
 *   closest hit       :=  argmin over ALL triangles of (t_eff, global triangle id)
 *   any hit           :=  exists tri: accept(tri, ray)
 * Because the floating-point slab test is monotone under box inclusion, every BVH whose node boxes
 * contain their triangles' AABBs exactly returns this same answer, whatever its shape or visit order.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef ORC_F64
/* liborc_f64.so (tests only): THIS source with every float a double -- arrays, records and arithmetic alike (constants keep their
 * float-rounded values).  It exists so that the restatement can be compared with the independent numpy one (tests/np_shading.py, fp64)
 * far below float rounding, which separates transcription errors (the same in both builds) from float32 conditioning (tests/test_oracle.py).
 * The defines come after the system headers and before art_oracle.h, whose API then takes doubles.  Packing / presentation (bit-level
 * float formats) are not part of this build. */
#define float double
#define fmaf fma
#define sqrtf sqrt
#define powf pow
#define acosf acos
#define tanf tan
#define fabsf fabs
#define fminf fmin
#define fmaxf fmax
#define floorf floor
#define copysignf copysign
#endif
#include "art_oracle.h"

/* ------------------------------------------------------------------ small vector maths */
typedef struct { float x, y, z; } v3;
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scl3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
    return V3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline float len3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 nrm3(v3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return scl3(a, inv); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }
/* column-major mat4 (m[c*4+r]) times (x,y,z,w): ((c0*x + c1*y) + c2*z) + c3*w */
static inline void mat4_mul4(const float *m, float x, float y, float z, float w, float out[4]) {
    for (int r = 0; r < 4; r++) out[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}
/* row-major 3x4 times point: ((m0*x + m1*y) + m2*z) + m3 */
static inline v3 xform_point(const float *m, v3 p) {
    return V3(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
              ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]);
}
static inline v3 xform_vec(const float *m, v3 p) {
    return V3((m[0] * p.x + m[1] * p.y) + m[2] * p.z, (m[4] * p.x + m[5] * p.y) + m[6] * p.z,
              (m[8] * p.x + m[9] * p.y) + m[10] * p.z);
}

/* ------------------------------------------------------------------ scene */
typedef struct {
    float *verts;     /* nv x 12 */
    uint32_t nv;
    uint32_t *idx;    /* widened to u32 */
    uint32_t n_idx, idx_bytes;
    uint8_t *tex;     /* 3 x th x tw x 4 */
    uint32_t tw, th;
    float o2w[12], w2o[12];
    uint32_t first_tri, n_tri;
} Prim;

struct OrcScene {
    Prim *prims;
    uint32_t n_prims, cap_prims;
    uint32_t T;
    /* flattened world-space soup */
    float *tv;          /* T x 9 */
    uint32_t *tri_prim; /* gid -> primitive */
    /* canonical LBVH */
    uint32_t *leaf_gid; /* sorted position -> gid */
    uint64_t *keys;
    int32_t *child;     /* 2*(T-1) */
    float *node_lo, *node_hi; /* (T-1) x 3 */
    float *leaf_lo, *leaf_hi; /* T x 3 (sorted position) */
    int built;
};

OrcScene *orc_scene_create(void) { return (OrcScene *)calloc(1, sizeof(OrcScene)); }

static void free_bvh(OrcScene *s) {
    free(s->tv); free(s->tri_prim); free(s->leaf_gid); free(s->keys); free(s->child);
    free(s->node_lo); free(s->node_hi); free(s->leaf_lo); free(s->leaf_hi);
    s->tv = NULL; s->tri_prim = NULL; s->leaf_gid = NULL; s->keys = NULL; s->child = NULL;
    s->node_lo = s->node_hi = s->leaf_lo = s->leaf_hi = NULL; s->built = 0;
}

void orc_scene_destroy(OrcScene *s) {
    if (!s) return;
    for (uint32_t i = 0; i < s->n_prims; i++) { free(s->prims[i].verts); free(s->prims[i].idx); free(s->prims[i].tex); }
    free(s->prims);
    free_bvh(s);
    free(s);
}

/* inverse of a row-major affine 3x4 by cofactors of the 3x3 part */
static void affine_inverse(const float *m, float *o) {
    float a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    float det = a * A + b * B + c * C;
    float id = 1.0f / det;
    o[0] = A * id;  o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
    o[4] = B * id;  o[5] = (a * i - c * g) * id;  o[6] = -(a * f - c * d) * id;
    o[8] = C * id;  o[9] = -(a * h - b * g) * id; o[10] = (a * e - b * d) * id;
    float tx = m[3], ty = m[7], tz = m[11];
    o[3] = -((o[0] * tx + o[1] * ty) + o[2] * tz);
    o[7] = -((o[4] * tx + o[5] * ty) + o[6] * tz);
    o[11] = -((o[8] * tx + o[9] * ty) + o[10] * tz);
}

int orc_scene_add_primitive(OrcScene *s, const float *verts, uint32_t nv, const void *idx, uint32_t n_idx,
                            uint32_t idx_bytes, const uint8_t *tex, uint32_t tw, uint32_t th, const float model3x4[12]) {
    if (!s || !verts || !idx || !tex || (idx_bytes != 2 && idx_bytes != 4) || n_idx % 3 || !tw || !th) return -1;
    if (s->n_prims == s->cap_prims) {
        s->cap_prims = s->cap_prims ? 2 * s->cap_prims : 16;
        s->prims = (Prim *)realloc(s->prims, s->cap_prims * sizeof(Prim));
    }
    Prim *p = &s->prims[s->n_prims];
    memset(p, 0, sizeof(*p));
    p->nv = nv; p->n_idx = n_idx; p->idx_bytes = idx_bytes; p->tw = tw; p->th = th;
    p->verts = (float *)malloc((size_t)nv * 12 * sizeof(float)); memcpy(p->verts, verts, (size_t)nv * 12 * sizeof(float));
    p->idx = (uint32_t *)malloc((size_t)n_idx * 4);
    for (uint32_t i = 0; i < n_idx; i++) {
        uint32_t v = idx_bytes == 2 ? ((const uint16_t *)idx)[i] : ((const uint32_t *)idx)[i];
        if (v >= nv) return -2;
        p->idx[i] = v;
    }
    size_t tb = (size_t)3 * tw * th * 4;
    p->tex = (uint8_t *)malloc(tb); memcpy(p->tex, tex, tb);
    memcpy(p->o2w, model3x4, 12 * sizeof(float));
    affine_inverse(p->o2w, p->w2o);
    p->n_tri = n_idx / 3;
    s->n_prims++;
    s->built = 0;
    return (int)(s->n_prims - 1);
}

uint32_t orc_scene_num_tris(const OrcScene *s) { return s->T; }

/* ------------------------------------------------------------------ canonical LBVH (Karras 2012) */
static inline uint32_t expand10(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
static inline uint64_t expand21(uint64_t v) {
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

typedef struct { uint64_t key; uint32_t gid; } KeyId;
static int cmp_keyid(const void *a, const void *b) {
    const KeyId *x = (const KeyId *)a, *y = (const KeyId *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->gid < y->gid ? -1 : (x->gid > y->gid ? 1 : 0);
}

/* common-prefix length of the (key, gid) bit strings at sorted positions i and j; -1 out of range */
static inline int delta_fn(const OrcScene *s, int i, int j) {
    if (j < 0 || j >= (int)s->T) return -1;
    uint64_t a = s->keys[i], b = s->keys[j];
    if (a != b) return __builtin_clzll(a ^ b);
    return 64 + __builtin_clz(s->leaf_gid[i] ^ s->leaf_gid[j]);
}

int orc_scene_build(OrcScene *s, int morton_bits) {
    if (!s || (morton_bits != 30 && morton_bits != 63)) return -1;
    free_bvh(s);
    uint32_t T = 0;
    for (uint32_t p = 0; p < s->n_prims; p++) { s->prims[p].first_tri = T; T += s->prims[p].n_tri; }
    s->T = T;
    if (T == 0) return -2;
    s->tv = (float *)malloc((size_t)T * 9 * sizeof(float));
    s->tri_prim = (uint32_t *)malloc((size_t)T * 4);
    float *tlo = (float *)malloc((size_t)T * 3 * sizeof(float)), *thi = (float *)malloc((size_t)T * 3 * sizeof(float));
    v3 cmin = V3(INFINITY, INFINITY, INFINITY), cmax = V3(-INFINITY, -INFINITY, -INFINITY);
    for (uint32_t p = 0; p < s->n_prims; p++) {
        const Prim *pr = &s->prims[p];
        for (uint32_t t = 0; t < pr->n_tri; t++) {
            uint32_t g = pr->first_tri + t;
            v3 w[3];
            for (int k = 0; k < 3; k++) {
                w[k] = xform_point(pr->o2w, ld3(pr->verts + (size_t)pr->idx[3 * t + k] * 12));
                s->tv[(size_t)g * 9 + 3 * k + 0] = w[k].x; s->tv[(size_t)g * 9 + 3 * k + 1] = w[k].y; s->tv[(size_t)g * 9 + 3 * k + 2] = w[k].z;
            }
            s->tri_prim[g] = p;
            v3 lo = V3(fminf(fminf(w[0].x, w[1].x), w[2].x), fminf(fminf(w[0].y, w[1].y), w[2].y), fminf(fminf(w[0].z, w[1].z), w[2].z));
            v3 hi = V3(fmaxf(fmaxf(w[0].x, w[1].x), w[2].x), fmaxf(fmaxf(w[0].y, w[1].y), w[2].y), fmaxf(fmaxf(w[0].z, w[1].z), w[2].z));
            tlo[3 * g] = lo.x; tlo[3 * g + 1] = lo.y; tlo[3 * g + 2] = lo.z;
            thi[3 * g] = hi.x; thi[3 * g + 1] = hi.y; thi[3 * g + 2] = hi.z;
            v3 c = scl3(add3(lo, hi), 0.5f);
            cmin = V3(fminf(cmin.x, c.x), fminf(cmin.y, c.y), fminf(cmin.z, c.z));
            cmax = V3(fmaxf(cmax.x, c.x), fmaxf(cmax.y, c.y), fmaxf(cmax.z, c.z));
        }
    }
    /* Morton keys of the AABB centroids, normalised over the centroid bounds */
    float cells = morton_bits == 30 ? 1024.0f : 2097152.0f;
    v3 ext = sub3(cmax, cmin);
    v3 sc = V3(ext.x > 0 ? cells / ext.x : 0.0f, ext.y > 0 ? cells / ext.y : 0.0f, ext.z > 0 ? cells / ext.z : 0.0f);
    KeyId *ki = (KeyId *)malloc((size_t)T * sizeof(KeyId));
    for (uint32_t g = 0; g < T; g++) {
        v3 c = scl3(add3(ld3(tlo + 3 * g), ld3(thi + 3 * g)), 0.5f);
        float qx = fminf(fmaxf((c.x - cmin.x) * sc.x, 0.0f), cells - 1.0f);
        float qy = fminf(fmaxf((c.y - cmin.y) * sc.y, 0.0f), cells - 1.0f);
        float qz = fminf(fmaxf((c.z - cmin.z) * sc.z, 0.0f), cells - 1.0f);
        uint64_t key;
        if (morton_bits == 30)
            key = ((uint64_t)expand10((uint32_t)qx) << 2) | ((uint64_t)expand10((uint32_t)qy) << 1) | (uint64_t)expand10((uint32_t)qz);
        else
            key = (expand21((uint64_t)qx) << 2) | (expand21((uint64_t)qy) << 1) | expand21((uint64_t)qz);
        ki[g].key = key; ki[g].gid = g;
    }
    qsort(ki, T, sizeof(KeyId), cmp_keyid);
    s->leaf_gid = (uint32_t *)malloc((size_t)T * 4);
    s->keys = (uint64_t *)malloc((size_t)T * 8);
    s->leaf_lo = (float *)malloc((size_t)T * 3 * sizeof(float));
    s->leaf_hi = (float *)malloc((size_t)T * 3 * sizeof(float));
    for (uint32_t i = 0; i < T; i++) {
        s->leaf_gid[i] = ki[i].gid; s->keys[i] = ki[i].key;
        memcpy(s->leaf_lo + 3 * i, tlo + 3 * ki[i].gid, 3 * sizeof(float));
        memcpy(s->leaf_hi + 3 * i, thi + 3 * ki[i].gid, 3 * sizeof(float));
    }
    free(ki); free(tlo); free(thi);
    uint32_t NI = T > 1 ? T - 1 : 0;
    s->child = (int32_t *)malloc((size_t)(NI ? NI : 1) * 8);
    s->node_lo = (float *)malloc((size_t)(NI ? NI : 1) * 3 * sizeof(float));
    s->node_hi = (float *)malloc((size_t)(NI ? NI : 1) * 3 * sizeof(float));
    int32_t *parent_int = (int32_t *)malloc((size_t)(NI ? NI : 1) * 4);
    int32_t *parent_leaf = (int32_t *)malloc((size_t)T * 4);
    if (NI) parent_int[0] = -1;
    for (int i = 0; i < (int)NI; i++) {
        int d = delta_fn(s, i, i + 1) - delta_fn(s, i, i - 1) >= 0 ? 1 : -1;
        int dmin = delta_fn(s, i, i - d);
        int lmax = 2;
        while (delta_fn(s, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (delta_fn(s, i, i + (l + t) * d) > dmin) l += t;
        int j = i + l * d;
        int dnode = delta_fn(s, i, j);
        int sp = 0, t = l;
        do {
            t = (t + 1) / 2;
            if (delta_fn(s, i, i + (sp + t) * d) > dnode) sp += t;
        } while (t > 1);
        int gamma = i + sp * d + (d < 0 ? -1 : 0);
        int lo = i < j ? i : j, hi = i < j ? j : i;
        if (lo == gamma) { s->child[2 * i] = ~gamma; parent_leaf[gamma] = i; }
        else { s->child[2 * i] = gamma; parent_int[gamma] = i; }
        if (hi == gamma + 1) { s->child[2 * i + 1] = ~(gamma + 1); parent_leaf[gamma + 1] = i; }
        else { s->child[2 * i + 1] = gamma + 1; parent_int[gamma + 1] = i; }
    }
    /* bottom-up refit: a node is finished by the second child to arrive */
    if (NI) {
        uint8_t *cnt = (uint8_t *)calloc(NI, 1);
        for (uint32_t lf = 0; lf < T; lf++) {
            int n = parent_leaf[lf];
            while (n >= 0) {
                if (++cnt[n] < 2) break;
                float lo[3], hi[3];
                for (int c = 0; c < 2; c++) {
                    int ch = s->child[2 * n + c];
                    const float *cl = ch < 0 ? s->leaf_lo + 3 * (~ch) : s->node_lo + 3 * ch;
                    const float *chh = ch < 0 ? s->leaf_hi + 3 * (~ch) : s->node_hi + 3 * ch;
                    for (int k = 0; k < 3; k++) {
                        lo[k] = c == 0 ? cl[k] : fminf(lo[k], cl[k]);
                        hi[k] = c == 0 ? chh[k] : fmaxf(hi[k], chh[k]);
                    }
                }
                memcpy(s->node_lo + 3 * n, lo, 3 * sizeof(float)); memcpy(s->node_hi + 3 * n, hi, 3 * sizeof(float));
                n = parent_int[n];
            }
        }
        free(cnt);
    }
    free(parent_int); free(parent_leaf);
    s->built = 1;
    return 0;
}

void orc_scene_get_lbvh(const OrcScene *s, uint32_t *leaf_gid, uint64_t *keys, int32_t *child, float *node_lo,
                        float *node_hi, float *leaf_lo, float *leaf_hi, float *tri_verts) {
    uint32_t T = s->T, NI = T > 1 ? T - 1 : 0;
    if (leaf_gid) memcpy(leaf_gid, s->leaf_gid, (size_t)T * 4);
    if (keys) memcpy(keys, s->keys, (size_t)T * 8);
    if (child) memcpy(child, s->child, (size_t)NI * 8);
    if (node_lo) memcpy(node_lo, s->node_lo, (size_t)NI * 3 * sizeof(float));
    if (node_hi) memcpy(node_hi, s->node_hi, (size_t)NI * 3 * sizeof(float));
    if (leaf_lo) memcpy(leaf_lo, s->leaf_lo, (size_t)T * 3 * sizeof(float));
    if (leaf_hi) memcpy(leaf_hi, s->leaf_hi, (size_t)T * 3 * sizeof(float));
    if (tri_verts) memcpy(tri_verts, s->tv, (size_t)T * 9 * sizeof(float));
}

/* ------------------------------------------------------------------ ray queries */
typedef struct {
    v3 o, d;
    float tmin, tmax;
    v3 inv, ood; /* slab precompute */
} Ray;

static inline float safe_dir(float d) { return fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d; }

static inline void ray_init(Ray *r, v3 o, v3 d, float tmin, float tmax) {
    r->o = o; r->d = d; r->tmin = tmin; r->tmax = tmax;
    r->inv = V3(1.0f / safe_dir(d.x), 1.0f / safe_dir(d.y), 1.0f / safe_dir(d.z));
    r->ood = V3(o.x * r->inv.x, o.y * r->inv.y, o.z * r->inv.z);
}

/* monotone slab test; returns pass/fail against [tmin, tlimit], writes the un-clamped entry distance */
static inline int slab(const Ray *r, const float *lo, const float *hi, float tlimit, float *tentry) {
    float t0x = fmaf(lo[0], r->inv.x, -r->ood.x), t1x = fmaf(hi[0], r->inv.x, -r->ood.x);
    float t0y = fmaf(lo[1], r->inv.y, -r->ood.y), t1y = fmaf(hi[1], r->inv.y, -r->ood.y);
    float t0z = fmaf(lo[2], r->inv.z, -r->ood.z), t1z = fmaf(hi[2], r->inv.z, -r->ood.z);
    float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    *tentry = tn;
    return fmaxf(tn, r->tmin) <= fminf(tf, tlimit);
}

/* Moller-Trumbore, two-sided (instance flags 0: vk_model.rs:374), tmin < t < tmax.
 * Edges are fattened by ORC_BARY_EPS in barycentric units: hardware ray tracing is watertight along shared
 * edges, plain Moller-Trumbore is not (a ray through a shared edge can miss both triangles by rounding). */
#define ORC_BARY_EPS 1.0e-6f
static inline int moller_trumbore(const Ray *r, const float *tv, float *t, float *u, float *v) {
    v3 v0 = ld3(tv), v1 = ld3(tv + 3), v2 = ld3(tv + 6);
    v3 e1 = sub3(v1, v0), e2 = sub3(v2, v0);
    v3 p = cross3(r->d, e2);
    float det = dot3(e1, p);
    if (det == 0.0f) return 0;
    float inv = 1.0f / det;
    v3 tvec = sub3(r->o, v0);
    float uu = dot3(tvec, p) * inv;
    if (!(uu >= -ORC_BARY_EPS && uu <= 1.0f + ORC_BARY_EPS)) return 0;
    v3 q = cross3(tvec, e1);
    float vv = dot3(r->d, q) * inv;
    if (!(vv >= -ORC_BARY_EPS && uu + vv <= 1.0f + ORC_BARY_EPS)) return 0;
    float tt = dot3(e2, q) * inv;
    if (!(tt > r->tmin && tt < r->tmax)) return 0;
    *t = tt; *u = uu; *v = vv;
    return 1;
}

typedef struct { float t, u, v; uint32_t gid; int hit; } Best;

static inline void consider_tri(const OrcScene *s, const Ray *r, uint32_t gid, const float *lo, const float *hi, Best *b) {
    float tn;
    if (!slab(r, lo, hi, b->t, &tn)) return;
    float t, u, v;
    if (!moller_trumbore(r, s->tv + (size_t)gid * 9, &t, &u, &v)) return;
    float te = fmaxf(t, tn);
    if (!b->hit || te < b->t || (te == b->t && gid < b->gid)) {
        b->t = te; b->u = u; b->v = v; b->gid = gid; b->hit = 1;
    }
}

static inline void tri_box(const float *tv, float *lo, float *hi) {
    for (int k = 0; k < 3; k++) {
        lo[k] = fminf(fminf(tv[k], tv[3 + k]), tv[6 + k]);
        hi[k] = fmaxf(fmaxf(tv[k], tv[3 + k]), tv[6 + k]);
    }
}

#define ORC_STACK 256

/* Packet-level accounting (bench.py's roofline): the GPU walks an 8x8 pixel block's rays as ONE packet that fetches every node and
 * triangle on the union of its rays' paths once.  A recorder, when a thread has one, marks what the rays of the current packet
 * touch; a node or triangle counts once per packet.  kind 0 = the primary packet, 1 + i = the shadow packet towards light i. */
typedef struct { uint32_t *stamp[17]; size_t words; uint32_t id; int kind; uint64_t nodes[2], tris[2]; } Visit;
static __thread Visit *g_visit = NULL;
static inline void visit_mark(size_t slot, int is_tri) {
    Visit *v = g_visit;
    uint32_t *st = v->stamp[v->kind];
    if (st[slot] != v->id) { st[slot] = v->id; if (is_tri) v->tris[v->kind != 0]++; else v->nodes[v->kind != 0]++; }
}

static Best closest_bvh(const OrcScene *s, const Ray *r, uint64_t *n_int, uint64_t *n_tri) {
    Best b; b.t = r->tmax; b.u = b.v = 0; b.gid = 0; b.hit = 0;
    uint32_t T = s->T;
    if (T == 1) {
        float tn;
        if (slab(r, s->leaf_lo, s->leaf_hi, b.t, &tn)) { (*n_tri)++; consider_tri(s, r, s->leaf_gid[0], s->leaf_lo, s->leaf_hi, &b); }
        return b;
    }
    int32_t stack[ORC_STACK]; float stack_t[ORC_STACK]; int sp = 0;
    stack[sp] = 0; stack_t[sp++] = -INFINITY;
    while (sp) {
        int32_t n = stack[--sp];
        if (fmaxf(stack_t[sp], r->tmin) > b.t) continue;
        (*n_int)++;
        if (g_visit) visit_mark((size_t)n, 0);
        int32_t c[2] = {s->child[2 * n], s->child[2 * n + 1]};
        float te[2]; int h[2];
        for (int k = 0; k < 2; k++) {
            const float *lo = c[k] < 0 ? s->leaf_lo + 3 * (~c[k]) : s->node_lo + 3 * c[k];
            const float *hi = c[k] < 0 ? s->leaf_hi + 3 * (~c[k]) : s->node_hi + 3 * c[k];
            h[k] = slab(r, lo, hi, b.t, &te[k]);
        }
        int first = te[0] <= te[1] ? 0 : 1;
        /* leaves are intersected at once, near one first; internal children are pushed far one first */
        for (int k = 0; k < 2; k++) {
            int ci = k == 0 ? first : 1 - first;
            if (!h[ci] || c[ci] >= 0) continue;
            if (fmaxf(te[ci], r->tmin) > b.t) continue;
            uint32_t pos = (uint32_t)~c[ci];
            (*n_tri)++;
            if (g_visit) visit_mark((size_t)(T - 1) + pos, 1);
            consider_tri(s, r, s->leaf_gid[pos], s->leaf_lo + 3 * pos, s->leaf_hi + 3 * pos, &b);
        }
        for (int k = 0; k < 2; k++) {
            int ci = k == 0 ? 1 - first : first;
            if (!h[ci] || c[ci] < 0) continue;
            if (sp >= ORC_STACK) abort();
            stack[sp] = c[ci]; stack_t[sp++] = te[ci];
        }
    }
    return b;
}

static Best closest_brute(const OrcScene *s, const Ray *r, uint64_t *n_tri) {
    Best b; b.t = r->tmax; b.u = b.v = 0; b.gid = 0; b.hit = 0;
    for (uint32_t g = 0; g < s->T; g++) {
        float lo[3], hi[3];
        tri_box(s->tv + (size_t)g * 9, lo, hi);
        (*n_tri)++;
        consider_tri(s, r, g, lo, hi, &b);
    }
    return b;
}

static inline int accept_any(const OrcScene *s, const Ray *r, uint32_t gid, const float *lo, const float *hi) {
    float tn, t, u, v;
    if (!slab(r, lo, hi, r->tmax, &tn)) return 0;
    return moller_trumbore(r, s->tv + (size_t)gid * 9, &t, &u, &v);
}

static int any_bvh(const OrcScene *s, const Ray *r, uint64_t *n_int, uint64_t *n_tri) {
    uint32_t T = s->T;
    if (T == 1) { (*n_tri)++; return accept_any(s, r, s->leaf_gid[0], s->leaf_lo, s->leaf_hi); }
    int32_t stack[ORC_STACK]; int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        int32_t n = stack[--sp];
        (*n_int)++;
        if (g_visit) visit_mark((size_t)n, 0);
        int32_t c[2] = {s->child[2 * n], s->child[2 * n + 1]};
        float te[2]; int h[2];
        for (int k = 0; k < 2; k++) {
            const float *lo = c[k] < 0 ? s->leaf_lo + 3 * (~c[k]) : s->node_lo + 3 * c[k];
            const float *hi = c[k] < 0 ? s->leaf_hi + 3 * (~c[k]) : s->node_hi + 3 * c[k];
            h[k] = slab(r, lo, hi, r->tmax, &te[k]);
        }
        int first = te[0] <= te[1] ? 0 : 1;
        for (int k = 0; k < 2; k++) {
            int ci = k == 0 ? first : 1 - first;
            if (!h[ci] || c[ci] >= 0) continue;
            uint32_t pos = (uint32_t)~c[ci];
            (*n_tri)++;
            if (g_visit) visit_mark((size_t)(T - 1) + pos, 1);
            if (accept_any(s, r, s->leaf_gid[pos], s->leaf_lo + 3 * pos, s->leaf_hi + 3 * pos)) return 1;
        }
        for (int k = 0; k < 2; k++) {
            int ci = k == 0 ? 1 - first : first;
            if (!h[ci] || c[ci] < 0) continue;
            if (sp >= ORC_STACK) abort();
            stack[sp++] = c[ci];
        }
    }
    return 0;
}

static int any_brute(const OrcScene *s, const Ray *r, uint64_t *n_tri) {
    for (uint32_t g = 0; g < s->T; g++) {
        float lo[3], hi[3];
        tri_box(s->tv + (size_t)g * 9, lo, hi);
        (*n_tri)++;
        if (accept_any(s, r, g, lo, hi)) return 1;
    }
    return 0;
}

void orc_trace_closest(const OrcScene *s, const float *rays, uint32_t n, int mode, float *out_tuv, int32_t *out_id,
                       uint64_t *n_int, uint64_t *n_tri) {
    uint64_t ni = 0, nt = 0;
    for (uint32_t i = 0; i < n; i++) {
        const float *q = rays + (size_t)i * 8;
        Ray r; ray_init(&r, ld3(q), ld3(q + 4), q[3], q[7]);
        Best b = mode == 0 ? closest_bvh(s, &r, &ni, &nt) : closest_brute(s, &r, &nt);
        if (b.hit) {
            uint32_t p = s->tri_prim[b.gid];
            out_tuv[4 * i] = b.t; out_tuv[4 * i + 1] = b.u; out_tuv[4 * i + 2] = b.v; out_tuv[4 * i + 3] = 0;
            out_id[2 * i] = (int32_t)p; out_id[2 * i + 1] = (int32_t)(b.gid - s->prims[p].first_tri);
        } else {
            out_tuv[4 * i] = q[7]; out_tuv[4 * i + 1] = out_tuv[4 * i + 2] = out_tuv[4 * i + 3] = 0;
            out_id[2 * i] = -1; out_id[2 * i + 1] = -1;
        }
    }
    if (n_int) *n_int = ni;
    if (n_tri) *n_tri = nt;
}

void orc_trace_any(const OrcScene *s, const float *rays, uint32_t n, int mode, uint8_t *out_hit, uint64_t *n_int,
                   uint64_t *n_tri) {
    uint64_t ni = 0, nt = 0;
    for (uint32_t i = 0; i < n; i++) {
        const float *q = rays + (size_t)i * 8;
        Ray r; ray_init(&r, ld3(q), ld3(q + 4), q[3], q[7]);
        out_hit[i] = (uint8_t)(mode == 0 ? any_bvh(s, &r, &ni, &nt) : any_brute(s, &r, &nt));
    }
    if (n_int) *n_int = ni;
    if (n_tri) *n_tri = nt;
}

/* ------------------------------------------------------------------ camera + lights (host maths) */
/* general 4x4 inverse by cofactors (nalgebra try_inverse stand-in; vk_camera.rs:111-113) */
static int mat4_inverse(const float *m, float *o) {
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.0f) return 0;
    det = 1.0f / det;
    for (int i = 0; i < 16; i++) o[i] = inv[i] * det;
    return 1;
}

void orc_camera_from_params(const float pos[3], const float dir[3], float aspect, float fovy, float znear, float zfar,
                            OrcCamera *out) {
    /* view = look_at_rh(pos, pos+dir, up=(0,-1,0))  (vk_camera.rs:182-189); dir is normalised by set_dir (:133-136) */
    v3 eye = ld3(pos);
    v3 f = nrm3(sub3(add3(eye, nrm3(ld3(dir))), eye));
    v3 up = V3(0.0f, -1.0f, 0.0f);
    v3 sv = nrm3(cross3(f, up));
    v3 u = cross3(sv, f);
    float *V = out->view;
    V[0] = sv.x; V[4] = sv.y; V[8] = sv.z;   V[12] = -dot3(sv, eye);
    V[1] = u.x;  V[5] = u.y;  V[9] = u.z;    V[13] = -dot3(u, eye);
    V[2] = -f.x; V[6] = -f.y; V[10] = -f.z;  V[14] = dot3(f, eye);
    V[3] = 0; V[7] = 0; V[11] = 0; V[15] = 1;
    /* proj = Perspective3::new(aspect, fovy, znear, zfar)  (vk_camera.rs:191-193), OpenGL-style z in [-1,1] */
    float *P = out->proj;
    memset(P, 0, 64);
    float c = 1.0f / tanf(fovy * 0.5f);
    P[0] = c / aspect; P[5] = c;
    P[10] = (zfar + znear) / (znear - zfar);
    P[14] = 2.0f * zfar * znear / (znear - zfar);
    P[11] = -1.0f;
    mat4_inverse(out->view, out->view_inv);
    mat4_inverse(out->proj, out->proj_inv);
    out->camera_pos[0] = pos[0]; out->camera_pos[1] = pos[1]; out->camera_pos[2] = pos[2];
}

static void light_zero(OrcLight *l) { memset(l, 0, sizeof(*l)); }
static void st3(float *d, const float *s) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }

void orc_light_point(const float pos[3], const float color[3], float falloff, int casts, OrcLight *o) { /* lights.rs:144-159 */
    light_zero(o); st3(o->pos, pos); o->type = 0; o->casts_shadows = casts ? 1u : 0u; st3(o->color, color); o->falloff_distance = falloff;
}
void orc_light_spot(const float pos[3], const float dir[3], const float color[3], float falloff, float penumbra,
                    float umbra, int casts, OrcLight *o) { /* lights.rs:228-243 */
    light_zero(o); st3(o->pos, pos); o->type = 1; st3(o->dir, dir); o->casts_shadows = casts ? 1u : 0u; st3(o->color, color);
    o->falloff_distance = falloff; o->penumbra_angle = penumbra; o->umbra_angle = umbra;
}
void orc_light_directional(const float dir[3], const float color[3], int casts, OrcLight *o) { /* lights.rs:281-296 */
    light_zero(o); o->type = 2; st3(o->dir, dir); o->casts_shadows = casts ? 1u : 0u; st3(o->color, color);
}
void orc_light_area(const float pos[3], const float pos2[3], const float pos3[3], int invert_normal, const float color[3],
                    float falloff, float penumbra, float umbra, int casts, OrcLight *o) { /* lights.rs:383-403 */
    light_zero(o);
    v3 n = cross3(sub3(ld3(pos), ld3(pos2)), sub3(ld3(pos3), ld3(pos2)));
    if (invert_normal) n = neg3(n);
    n = nrm3(n);
    st3(o->pos, pos); o->type = 3; o->dir[0] = n.x; o->dir[1] = n.y; o->dir[2] = n.z; o->casts_shadows = casts ? 1u : 0u;
    st3(o->color, color); o->falloff_distance = falloff; st3(o->area_pos2, pos2); o->penumbra_angle = penumbra;
    st3(o->area_pos3, pos3); o->umbra_angle = umbra;
}

/* ------------------------------------------------------------------ ray generation (raytrace.rgen.glsl:78-88) */
static inline void primary_ray(const OrcCamera *cam, uint32_t x, uint32_t y, uint32_t w, uint32_t h, v3 *o, v3 *d) {
    float px = (float)x + 0.5f, py = (float)y + 0.5f;
    float ux = px / (float)w, uy = py / (float)h;
    float dx = ux * 2.0f - 1.0f, dy = uy * 2.0f - 1.0f;
    float org[4], tgt[4], dir[4];
    mat4_mul4(cam->view_inv, 0.0f, 0.0f, 0.0f, 1.0f, org);
    mat4_mul4(cam->proj_inv, dx, dy, 1.0f, 1.0f, tgt);
    v3 tn = nrm3(V3(tgt[0], tgt[1], tgt[2]));
    mat4_mul4(cam->view_inv, tn.x, tn.y, tn.z, 0.0f, dir);
    *o = V3(org[0], org[1], org[2]);
    *d = V3(dir[0], dir[1], dir[2]);
}

void orc_gen_primary(const OrcCamera *cam, uint32_t w, uint32_t h, float *rays) {
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            v3 o, d; primary_ray(cam, x, y, w, h, &o, &d);
            float *q = rays + ((size_t)y * w + x) * 8;
            q[0] = o.x; q[1] = o.y; q[2] = o.z; q[3] = 0.001f; q[4] = d.x; q[5] = d.y; q[6] = d.z; q[7] = 10000.0f;
        }
}

/* ------------------------------------------------------------------ lights (light.glsl) */
static v3 compute_barycentric(v3 a, v3 b, v3 c, v3 p) { /* light.glsl:50-68 */
    v3 v0 = sub3(b, a), v1 = sub3(c, a), v2 = sub3(p, a);
    float d00 = dot3(v0, v0), d01 = dot3(v0, v1), d11 = dot3(v1, v1), d20 = dot3(v2, v0), d21 = dot3(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    v3 r;
    r.x = (d11 * d20 - d01 * d21) / denom;
    r.y = (d00 * d21 - d01 * d20) / denom;
    r.z = 1.0f - r.x - r.y;
    return r;
}
static v3 closest_point_to_segment(v3 p0, v3 p1, v3 p) { /* light.glsl:70-75 */
    v3 v01 = sub3(p1, p0);
    float t = dot3(sub3(p, p0), v01) / dot3(v01, v01);
    t = clampf(t, 0.0f, 1.0f);
    return add3(p0, scl3(v01, t));
}
static v3 closest_point_to_triangle(v3 p0, v3 p1, v3 p2, v3 pt) { /* light.glsl:77-91 */
    v3 b = compute_barycentric(p0, p1, p2, pt);
    if (b.x < 0.0f) return closest_point_to_segment(p2, p0, pt);
    else if (b.z < 0.0f) return closest_point_to_segment(p1, p2, pt);
    return pt;
}
static v3 get_unnormalized_L_vec(const OrcLight *l, v3 pos) { /* light.glsl:93-124 */
    if (l->type == 0 || l->type == 1) return sub3(ld3(l->pos), pos);
    if (l->type == 2) return scl3(neg3(ld3(l->dir)), 10.0f);
    if (l->type == 3) {
        v3 ldir = ld3(l->dir), lp = ld3(l->pos), p2 = ld3(l->area_pos2), p3 = ld3(l->area_pos3);
        float distance = dot3(ldir, p2) - dot3(ldir, pos);
        v3 cp = add3(pos, scl3(ldir, distance));
        v3 b = compute_barycentric(lp, p2, p3, cp);
        v3 c;
        if (b.x < 0.0f) {
            v3 p4 = add3(sub3(lp, p2), p3);
            c = closest_point_to_triangle(lp, p3, p4, cp);
        } else if (b.y < 0.0f) c = closest_point_to_segment(lp, p2, cp);
        else if (b.z < 0.0f) c = closest_point_to_segment(p2, p3, cp);
        else c = cp;
        return sub3(c, pos);
    }
    return V3(1.0f, 1.0f, 1.0f);
}
static v3 get_light_radiance(const OrcLight *l, v3 pos, v3 L) { /* light.glsl:34-48 */
    v3 rad = ld3(l->color);
    if (l->type == 1 || l->type == 3) {
        float theta_s = acosf(clampf(dot3(ld3(l->dir), neg3(L)), -1.0f, 1.0f)); /* clamp: SURVEY appendix A */
        float t = clampf((theta_s - l->umbra_angle) / (l->penumbra_angle - l->umbra_angle), 0.0f, 1.0f);
        rad = scl3(rad, powf(t, 2.0f));
    }
    if (l->falloff_distance > 0.0f) {
        float dist = len3(sub3(ld3(l->pos), pos));
        rad = scl3(rad, powf(fmaxf(1.0f - powf(dist / l->falloff_distance, 2.0f), 0.0f), 2.0f));
    }
    return rad;
}

void orc_light_eval(const OrcLight *l, const float p[3], float nn_L[3], float radiance[3]) {
    v3 nl = get_unnormalized_L_vec(l, ld3(p));
    v3 r = get_light_radiance(l, ld3(p), nrm3(nl));
    nn_L[0] = nl.x; nn_L[1] = nl.y; nn_L[2] = nl.z;
    radiance[0] = r.x; radiance[1] = r.y; radiance[2] = r.z;
}

/* ------------------------------------------------------------------ BRDFs (brdfs.glsl) */
#define ORC_INV_PI (1.0f / 3.14159265359f)
static inline float D_GGX(float a_, float NdotH) { /* brdfs.glsl:6-14 */
    float om = 1.0f - NdotH * NdotH;
    float a = NdotH * a_;
    float k = a_ / (om + a * a);
    return k * k * ORC_INV_PI;
}
static inline float V_SmithGGXCorrelated_fast(float a_, float NdotV, float NdotL) { /* brdfs.glsl:25-29 */
    return 0.5f / mixf(2.0f * NdotL * NdotV, NdotL + NdotV, a_);
}
static inline float pow5(float x) { return powf(x, 5.0f); }
static inline float F_Schlick1(float F0, float F90, float x) { return F0 + (F90 - F0) * pow5(1.0f - x); } /* :44-49 */
static inline float Burley_diffuse_local_sss(float a_, float NdotV, float nc_NdotV, float nc_NdotL, float LdotH, float ratio) { /* :89-99 */
    float F_SS90 = a_ * LdotH * LdotH;
    float F_SS = F_Schlick1(1.0f, F_SS90, nc_NdotL) * F_Schlick1(1.0f, F_SS90, nc_NdotV);
    float f_ss = (1.0f / (nc_NdotV * nc_NdotL) - 0.5f) * F_SS + 0.5f;
    float local_sss = 1.25f * ratio * f_ss;
    float f90 = 0.5f + 2.0f * F_SS90;
    float diffuse = (1.0f - ratio) * F_Schlick1(1.0f, f90, nc_NdotL) * F_Schlick1(1.0f, f90, nc_NdotV);
    return NdotV * (diffuse + local_sss) * ORC_INV_PI;
}
void orc_brdf_terms(float NdotL, float NdotV, float NdotH, float LdotH, float nc_NdotV, float nc_NdotL, float alpha, float out[4]) {
    out[0] = D_GGX(alpha, NdotH);
    out[1] = V_SmithGGXCorrelated_fast(alpha, NdotV, NdotL);
    out[2] = pow5(1.0f - LdotH);
    out[3] = Burley_diffuse_local_sss(alpha, NdotV, nc_NdotV, nc_NdotL, LdotH, 0.4f);
}

/* ------------------------------------------------------------------ textures: linear / REPEAT / LOD 0 (vk_rt_descriptor_set.rs:42-56) */
static inline int wrapi(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
static void sample_tex(const Prim *p, int layer, float u, float v, float out[4]) {
    int tw = (int)p->tw, th = (int)p->th;
    float x = u * (float)tw - 0.5f, y = v * (float)th - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = wrapi((int)x0f, tw), y0 = wrapi((int)y0f, th);
    int x1 = wrapi(x0 + 1, tw), y1 = wrapi(y0 + 1, th);
    const uint8_t *base = p->tex + (size_t)layer * tw * th * 4;
    const uint8_t *t00 = base + ((size_t)y0 * tw + x0) * 4, *t10 = base + ((size_t)y0 * tw + x1) * 4;
    const uint8_t *t01 = base + ((size_t)y1 * tw + x0) * 4, *t11 = base + ((size_t)y1 * tw + x1) * 4;
    const float k = 1.0f / 255.0f;
    for (int c = 0; c < 4; c++) {
        float a = (float)t00[c] * k, b = (float)t10[c] * k, cc = (float)t01[c] * k, d = (float)t11[c] * k;
        float top = a * (1.0f - fx) + b * fx, bot = cc * (1.0f - fx) + d * fx;
        out[c] = top * (1.0f - fy) + bot * fy;
    }
}

/* ------------------------------------------------------------------ full pixel (raytrace.rgen.glsl:77-200) */
typedef struct {
    const OrcScene *s; const OrcCamera *cam; const OrcLight *lights; uint32_t nl, w, h, y0, y1;
    float *color, *depth, *normal, *hit_tuv; int32_t *hit_id; uint32_t *shadow_bits;
    OrcStats stats; pthread_mutex_t *mu; volatile uint32_t *next_row;
} Job;

static void render_pixel(Job *J, OrcStats *st, uint32_t x, uint32_t y) {
    const OrcScene *s = J->s; const OrcCamera *cam = J->cam;
    size_t pix = (size_t)y * J->w + x;
    v3 o, d; primary_ray(cam, x, y, J->w, J->h, &o, &d);
    Ray r; ray_init(&r, o, d, 0.001f, 10000.0f);
    st->primary_rays++;
    if (g_visit) g_visit->kind = 0;
    Best b = closest_bvh(s, &r, &st->n_int_primary, &st->n_tri_primary);
    float out_depth = 10000.0f;
    v3 out_color = V3(0, 0, 0), out_normal = V3(0.5f, 0.5f, 0.5f);
    uint32_t sbits = 0;
    if (J->hit_tuv) { float *q = J->hit_tuv + pix * 4; q[0] = b.hit ? b.t : 10000.0f; q[1] = b.u; q[2] = b.v; q[3] = 0; }
    if (J->hit_id) { J->hit_id[pix * 2] = -1; J->hit_id[pix * 2 + 1] = -1; }
    if (b.hit) {
        st->hit_pixels++;
        uint32_t pi = s->tri_prim[b.gid];
        const Prim *p = &s->prims[pi];
        uint32_t tri = b.gid - p->first_tri;
        if (J->hit_id) { J->hit_id[pix * 2] = (int32_t)pi; J->hit_id[pix * 2 + 1] = (int32_t)tri; }
        const float *a0 = p->verts + (size_t)p->idx[3 * tri] * 12, *a1 = p->verts + (size_t)p->idx[3 * tri + 1] * 12,
                    *a2 = p->verts + (size_t)p->idx[3 * tri + 2] * 12;
        float bx = 1.0f - b.u - b.v, by = b.u, bz = b.v;
        v3 pos = add3(add3(scl3(ld3(a0), bx), scl3(ld3(a1), by)), scl3(ld3(a2), bz));
        v3 world_pos = xform_point(p->o2w, pos);
        float tu = (a0[3] * bx + a1[3] * by) + a2[3] * bz, tvv = (a0[4] * bx + a1[4] * by) + a2[4] * bz;
        v3 nrm = nrm3(add3(add3(scl3(ld3(a0 + 5), bx), scl3(ld3(a1 + 5), by)), scl3(ld3(a2 + 5), bz)));
        /* normal * world_to_object: component c = dot(normal, column c of the 3x3 part) */
        const float *W = p->w2o;
        v3 world_normal = nrm3(V3(dot3(nrm, V3(W[0], W[4], W[8])), dot3(nrm, V3(W[1], W[5], W[9])), dot3(nrm, V3(W[2], W[6], W[10]))));
        v3 tan = nrm3(add3(add3(scl3(ld3(a0 + 8), bx), scl3(ld3(a1 + 8), by)), scl3(ld3(a2 + 8), bz)));
        v3 world_tangent = nrm3(xform_vec(p->o2w, tan));
        world_tangent = nrm3(sub3(world_tangent, scl3(world_normal, dot3(world_tangent, world_normal))));
        v3 world_binormal = scl3(cross3(world_normal, world_tangent), a0[11]);
        float tx[4];
        sample_tex(p, 2, tu, tvv, tx);
        v3 N = nrm3(V3(tx[0] * 2.0f - 1.0f, tx[1] * 2.0f - 1.0f, tx[2] * 2.0f - 1.0f));
        N = nrm3(add3(add3(scl3(world_tangent, N.x), scl3(world_binormal, N.y)), scl3(world_normal, N.z)));
        sample_tex(p, 0, tu, tvv, tx);
        v3 albedo = V3(powf(tx[0], 2.2f), powf(tx[1], 2.2f), powf(tx[2], 2.2f));
        sample_tex(p, 1, tu, tvv, tx);
        float roughness = tx[1], metallic = tx[2];
        v3 Vv = nrm3(sub3(ld3(cam->camera_pos), world_pos));
        v3 F0 = V3(mixf(0.04f, albedo.x, metallic), mixf(0.04f, albedo.y, metallic), mixf(0.04f, albedo.z, metallic));
        float alpha = roughness * roughness;
        float nc_NdotV = dot3(N, Vv);
        float NdotV = clampf(nc_NdotV, 1e-5f, 1.0f);
        v3 rho = V3(0, 0, 0);
        for (uint32_t i = 0; i < J->nl; i++) {
            const OrcLight *l = &J->lights[i];
            v3 nn_L = get_unnormalized_L_vec(l, world_pos);
            v3 L = nrm3(nn_L);
            v3 H = nrm3(add3(Vv, L));
            float nc_NdotL = dot3(N, L);
            float NdotL = clampf(nc_NdotL, 0.0f, 1.0f);
            float NdotH = clampf(dot3(N, H), 0.0f, 1.0f);
            float LdotH = clampf(dot3(L, H), 0.0f, 1.0f);
            float sch = pow5(1.0f - LdotH);
            v3 Ks = V3(F0.x + (1.0f - F0.x) * sch, F0.y + (1.0f - F0.y) * sch, F0.z + (1.0f - F0.z) * sch);
            v3 Kd = scl3(albedo, 1.0f - metallic);
            float DG = D_GGX(alpha, NdotH) * V_SmithGGXCorrelated_fast(alpha, NdotV, NdotL);
            v3 rho_s = scl3(Ks, DG);
            v3 rho_d = scl3(Kd, Burley_diffuse_local_sss(alpha, NdotV, nc_NdotV, nc_NdotL, LdotH, 0.4f));
            float att = 1.0f;
            if (l->casts_shadows && nc_NdotL > 0.0f) {
                Ray sr; ray_init(&sr, world_pos, L, 0.01f, len3(nn_L));
                st->shadow_rays++;
                if (i < 16) sbits |= 1u << (16 + i);
                if (g_visit) g_visit->kind = 1 + (int)(i < 16 ? i : 15);
                if (any_bvh(s, &sr, &st->n_int_shadow, &st->n_tri_shadow)) { att = 0.05f; if (i < 16) sbits |= 1u << i; }
            }
            v3 rad = get_light_radiance(l, world_pos, L);
            rho = add3(rho, scl3(scl3(mul3(add3(rho_s, rho_d), rad), att), NdotL));
        }
        out_color = rho;
        float vp[4];
        mat4_mul4(cam->view, world_pos.x, world_pos.y, world_pos.z, 1.0f, vp);
        out_depth = -vp[2];
        /* mat3(transpose(view_inv)) * N : component r = column r of view_inv (rows 0..2) . N */
        const float *VI = cam->view_inv;
        v3 on = V3((VI[0] * N.x + VI[1] * N.y) + VI[2] * N.z, (VI[4] * N.x + VI[5] * N.y) + VI[6] * N.z, (VI[8] * N.x + VI[9] * N.y) + VI[10] * N.z);
        on.y = -on.y; on.z = -on.z;
        on = nrm3(on);
        out_normal = V3(on.x * 0.5f + 0.5f, on.y * 0.5f + 0.5f, on.z * 0.5f + 0.5f);
    }
    if (!(isfinite(out_color.x) && isfinite(out_color.y) && isfinite(out_color.z))) st->nonfinite_pixels++;
    if (!J->color) return;   /* counting runs (orc_packet_stats) keep no image */
    float *c = J->color + pix * 4; c[0] = out_color.x; c[1] = out_color.y; c[2] = out_color.z; c[3] = 1.0f;
    J->depth[pix] = out_depth;
    float *n = J->normal + pix * 4; n[0] = out_normal.x; n[1] = out_normal.y; n[2] = out_normal.z; n[3] = 1.0f;
    if (J->shadow_bits) J->shadow_bits[pix] = sbits;
}

static void *worker(void *arg) {
    Job *J = (Job *)arg;
    OrcStats st; memset(&st, 0, sizeof(st));
    /* work items: 32 x 32-pixel tiles of the rows [y0, y1), dealt by an atomic cursor ("all host cores over 32x32-pixel tiles") */
    const uint32_t tiles_x = (J->w + 31u) / 32u, tiles_y = (J->y1 > J->y0 ? J->y1 - J->y0 + 31u : 0u) / 32u;
    for (;;) {
        uint32_t t = __sync_fetch_and_add(J->next_row, 1u);
        if (t >= tiles_x * tiles_y) break;
        const uint32_t xa = (t % tiles_x) * 32u, ya = J->y0 + (t / tiles_x) * 32u;
        const uint32_t xe = xa + 32u < J->w ? xa + 32u : J->w, ye = ya + 32u < J->y1 ? ya + 32u : J->y1;
        for (uint32_t yy = ya; yy < ye; yy++)
            for (uint32_t x = xa; x < xe; x++) render_pixel(J, &st, x, yy);
    }
    pthread_mutex_lock(J->mu);
    uint64_t *a = (uint64_t *)&J->stats, *b = (uint64_t *)&st;
    for (size_t i = 0; i < sizeof(OrcStats) / 8; i++) a[i] += b[i];
    pthread_mutex_unlock(J->mu);
    return NULL;
}

void orc_render(const OrcScene *s, const OrcCamera *cam, const OrcLight *lights, uint32_t n_lights, uint32_t w, uint32_t h,
                uint32_t y0, uint32_t y1, float *color, float *depth, float *normal, float *hit_tuv, int32_t *hit_id,
                uint32_t *shadow_bits, OrcStats *stats, int n_threads) {
    Job J; memset(&J, 0, sizeof(J));
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    volatile uint32_t next = 0;   /* the tile cursor (next_row of the packet statistics: a row of blocks) */
    J.s = s; J.cam = cam; J.lights = lights; J.nl = n_lights; J.w = w; J.h = h; J.y0 = y0; J.y1 = y1 < h ? y1 : h;
    J.color = color; J.depth = depth; J.normal = normal; J.hit_tuv = hit_tuv; J.hit_id = hit_id; J.shadow_bits = shadow_bits;
    J.mu = &mu; J.next_row = &next;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if (n_threads == 1) worker(&J);
    else {
        pthread_t th[256];
        for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, worker, &J);
        for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    }
    if (stats) *stats = J.stats;
}

/* ------------------------------------------------------------------ packet-level visit counts (bench.py roofline) */
typedef struct { Job J; uint32_t bw, bh; uint64_t out[4]; } PacketJob;
static void *packet_worker(void *arg) {
    PacketJob *P = (PacketJob *)arg; Job *J = &P->J;
    const uint32_t T = J->s->T, bx_n = (J->w + P->bw - 1) / P->bw;
    Visit v; memset(&v, 0, sizeof(v));
    v.words = (size_t)2 * T;
    for (uint32_t k = 0; k < 1 + (J->nl < 16 ? J->nl : 16); k++) v.stamp[k] = (uint32_t *)calloc(v.words, 4);
    g_visit = &v;
    OrcStats st; memset(&st, 0, sizeof(st));
    for (;;) {
        uint32_t by = __sync_fetch_and_add(J->next_row, 1u);          /* a row of blocks */
        if (by * P->bh >= J->h) break;
        for (uint32_t bx = 0; bx < bx_n; bx++) {
            v.id = by * bx_n + bx + 1;                                  /* never 0: the stamps start cleared */
            for (uint32_t y = by * P->bh; y < (by + 1) * P->bh && y < J->h; y++)
                for (uint32_t x = bx * P->bw; x < (bx + 1) * P->bw && x < J->w; x++) render_pixel(J, &st, x, y);
        }
    }
    g_visit = NULL;
    for (int k = 0; k < 17; k++) free(v.stamp[k]);
    pthread_mutex_lock(J->mu);
    P->out[0] += v.nodes[0]; P->out[1] += v.tris[0]; P->out[2] += v.nodes[1]; P->out[3] += v.tris[1];
    uint64_t *a = (uint64_t *)&J->stats, *b = (uint64_t *)&st;
    for (size_t i = 0; i < sizeof(OrcStats) / 8; i++) a[i] += b[i];
    pthread_mutex_unlock(J->mu);
    return NULL;
}
void orc_packet_stats(const OrcScene *s, const OrcCamera *cam, const OrcLight *lights, uint32_t n_lights, uint32_t w, uint32_t h,
                      uint32_t block_w, uint32_t block_h, uint64_t out[4], OrcStats *stats, int n_threads) {
    PacketJob P; memset(&P, 0, sizeof(P));
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    volatile uint32_t next = 0;
    P.J.s = s; P.J.cam = cam; P.J.lights = lights; P.J.nl = n_lights; P.J.w = w; P.J.h = h; P.J.y0 = 0; P.J.y1 = h;
    P.J.mu = &mu; P.J.next_row = &next; P.bw = block_w ? block_w : 8; P.bh = block_h ? block_h : 8;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if (s->T < 2) n_threads = 1;
    pthread_t th[256];
    for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, packet_worker, &P);
    for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    for (int k = 0; k < 4; k++) out[k] = P.out[k];
    if (stats) *stats = P.J.stats;
}

/* ------------------------------------------------------------------ ray-traced AO (BASELINE config 5) */
static inline uint32_t hilbert_index(uint32_t x, uint32_t y) { /* XeGTAO.h:120-142, XE_HILBERT_LEVEL 6 */
    uint32_t index = 0;
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
/* cos/sin of u turns (u in [0,1)) by quadrant reduction + Taylor polynomials in a fixed fmaf order: the same bits on any IEEE machine */
static inline void sincos_turns(float u, float *c, float *s) {
    float q = u * 4.0f;
    int k = (int)q;
    float x = (q - (float)k) * 1.57079632679489662f, x2 = x * x;
    float sp = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, -2.50521083854417188e-8f, 2.75573192239858907e-6f), -1.98412698412698413e-4f), 8.33333333333333333e-3f), -1.66666666666666667e-1f), 1.0f) * x;
    float cp = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 2.08767569878680990e-9f, -2.75573192239858907e-7f), 2.48015873015873016e-5f), -1.38888888888888889e-3f), 4.16666666666666667e-2f), -0.5f), 1.0f);
    switch (k & 3) {
        case 0: *c = cp; *s = sp; break;
        case 1: *c = -sp; *s = cp; break;
        case 2: *c = -cp; *s = -sp; break;
        default: *c = sp; *s = -cp; break;
    }
}

typedef struct {
    const OrcScene *s; const OrcCamera *cam; uint32_t w, h, spp; float radius; const float *depth, *normal; uint32_t *out; uint32_t lut[65];
    volatile uint32_t *next_row; pthread_mutex_t *mu; uint64_t n_rays, n_int, n_tri;
} AoJob;

static void *ao_worker(void *arg) {
    AoJob *J = (AoJob *)arg;
    uint64_t nr = 0, ni = 0, nt = 0;
    for (;;) {
        uint32_t y = __sync_fetch_and_add(J->next_row, 1u);
        if (y >= J->h) break;
        for (uint32_t x = 0; x < J->w; x++) {
            size_t pix = (size_t)y * J->w + x;
            float depth = J->depth[pix];
            if (!(depth < 10000.0f)) { J->out[pix] = 255u; continue; }
            /* position from the depth output: the primary ray scaled to view depth */
            float px = (float)x + 0.5f, py = (float)y + 0.5f;
            float dx = (px / (float)J->w) * 2.0f - 1.0f, dy = (py / (float)J->h) * 2.0f - 1.0f;
            float org[4], tg[4], dir[4];
            mat4_mul4(J->cam->view_inv, 0.0f, 0.0f, 0.0f, 1.0f, org);
            mat4_mul4(J->cam->proj_inv, dx, dy, 1.0f, 1.0f, tg);
            v3 tn = nrm3(V3(tg[0], tg[1], tg[2]));
            mat4_mul4(J->cam->view_inv, tn.x, tn.y, tn.z, 0.0f, dir);
            float sc = depth / -tn.z;
            v3 wp = V3(org[0] + dir[0] * sc, org[1] + dir[1] * sc, org[2] + dir[2] * sc);
            /* world normal from the view-space normal output (raytrace.rgen.glsl:192-194 inverted) */
            const float *nm = J->normal + pix * 4;
            float nx = nm[0] * 2.0f - 1.0f, ny = -(nm[1] * 2.0f - 1.0f), nz = -(nm[2] * 2.0f - 1.0f);
            const float *VI = J->cam->view_inv;
            v3 N = nrm3(V3((VI[0] * nx + VI[4] * ny) + VI[8] * nz, (VI[1] * nx + VI[5] * ny) + VI[9] * nz, (VI[2] * nx + VI[6] * ny) + VI[10] * nz));
            /* branchless orthonormal basis (Duff et al. 2017) */
            float sg = copysignf(1.0f, N.z), a = -1.0f / (sg + N.z), b = N.x * N.y * a;
            v3 T = V3(1.0f + sg * N.x * N.x * a, sg * b, -sg * N.x), B = V3(b, sg + N.y * N.y * a, -N.y);
            uint32_t hidx = hilbert_index(x & 63u, y & 63u), occ = 0;
            for (uint32_t sidx = 0; sidx < J->spp; sidx++) {
                float fi = (float)(hidx + 288u * sidx);
                float v1 = 0.5f + fi * 0.75487766624669276f, v2 = 0.5f + fi * 0.56984029099805327f;
                float u1 = v1 - floorf(v1), u2 = v2 - floorf(v2);
                float r = sqrtf(u1), cz = sqrtf(1.0f - u1), cc, ss;
                sincos_turns(u2, &cc, &ss);
                float lx = r * cc, ly = r * ss;
                v3 d = add3(add3(scl3(T, lx), scl3(B, ly)), scl3(N, cz));
                Ray ray; ray_init(&ray, wp, d, J->radius * 0.01f, J->radius);
                nr++;
                occ += (uint32_t)any_bvh(J->s, &ray, &ni, &nt);
            }
            J->out[pix] = J->lut[occ];
        }
    }
    pthread_mutex_lock(J->mu);
    J->n_rays += nr; J->n_int += ni; J->n_tri += nt;
    pthread_mutex_unlock(J->mu);
    return NULL;
}

void orc_render_ao(const OrcScene *s, const OrcCamera *cam, uint32_t w, uint32_t h, const float *depth, const float *normal, uint32_t spp,
                   float radius, uint32_t *out_ao, uint64_t *n_rays, uint64_t *n_int, uint64_t *n_tri, int n_threads) {
    AoJob J; memset(&J, 0, sizeof(J));
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    volatile uint32_t next = 0;
    if (spp > 64) spp = 64;
    J.s = s; J.cam = cam; J.w = w; J.h = h; J.spp = spp; J.radius = radius; J.depth = depth; J.normal = normal; J.out = out_ao; J.next_row = &next; J.mu = &mu;
    for (uint32_t k = 0; k <= spp; k++) J.lut[k] = (uint32_t)(pow(1.0 - (double)k / (double)spp, 2.2) * 255.0 + 0.5); /* XE_GTAO_DEFAULT_FINAL_VALUE_POWER */
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if (n_threads == 1) ao_worker(&J);
    else {
        pthread_t th[256];
        for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, ao_worker, &J);
        for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    }
    if (n_rays) *n_rays = J.n_rays;
    if (n_int) *n_int = J.n_int;
    if (n_tri) *n_tri = J.n_tri;
}

#ifndef ORC_F64 /* bit-level float formats: float build only */
/* ------------------------------------------------------------------ output packing + LPM tonemap */
/* float32 -> unsigned small float with 5 exponent bits and `mb` mantissa bits, round to nearest even; negatives -> 0, overflow -> +Inf */
static uint32_t pack_ufloat(float f, int mb) {
    uint32_t u; memcpy(&u, &f, 4);
    uint32_t e8 = (u >> 23) & 255u, m = u & 0x7FFFFFu;
    if (e8 == 255u && m) return (31u << mb) | 1u;               /* NaN */
    if (u >> 31) return 0;                                       /* negative (incl. -0, -Inf) */
    if (e8 == 255u) return 31u << mb;                            /* +Inf */
    int e = (int)e8 - 127 + 15;
    if (e >= 31) return 31u << mb;
    int shift = 23 - mb;
    uint32_t full = m | (e8 ? 0x800000u : 0u);
    if (e <= 0) { shift += 1 - e; e = 0; if (shift > 31) return 0; }
    else full &= 0x7FFFFFu;
    uint32_t q = full >> shift, rem = full & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    uint32_t out = ((uint32_t)e << mb) + q;                      /* a mantissa carry rolls into the exponent */
    return out > (31u << mb) ? (31u << mb) : out;
}
static float unpack_ufloat(uint32_t v, int mb) {
    uint32_t e = v >> mb, m = v & ((1u << mb) - 1u);
    if (e == 31u) return m ? NAN : INFINITY;
    if (e == 0) return ldexpf((float)m, -14 - mb);
    return ldexpf((float)(m | (1u << mb)), (int)e - 15 - mb);
}
uint32_t orc_pack_b10g11r11(const float rgb[3]) { return pack_ufloat(rgb[0], 6) | (pack_ufloat(rgb[1], 6) << 11) | (pack_ufloat(rgb[2], 5) << 22); }
void orc_unpack_b10g11r11(uint32_t v, float rgb[3]) { rgb[0] = unpack_ufloat(v & 0x7FFu, 6); rgb[1] = unpack_ufloat((v >> 11) & 0x7FFu, 6); rgb[2] = unpack_ufloat(v >> 22, 5); }
uint16_t orc_pack_f16(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    uint32_t sign = (u >> 16) & 0x8000u, e8 = (u >> 23) & 255u, m = u & 0x7FFFFFu;
    if (e8 == 255u) return (uint16_t)(sign | 0x7C00u | (m ? 0x200u : 0u));
    int e = (int)e8 - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    int shift = 13; uint32_t full = m | (e8 ? 0x800000u : 0u);
    if (e <= 0) { shift += 1 - e; e = 0; if (shift > 31) return (uint16_t)sign; } else full &= 0x7FFFFFu;
    uint32_t q = full >> shift, rem = full & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | (((uint32_t)e << 10) + q));
}

/* LpmColRgbToXyz with the reference's LpmColXyToZ (vk_tonemap.rs:12-47; z = 1 - x + y as written there) */
static void mat3_inverse(const float m[9], float o[9]) { /* row-major */
    float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g, det = a * A + b * B + c * C, id = 1.0f / det;
    o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
    o[3] = B * id; o[4] = (a * i - c * g) * id; o[5] = -(a * f - c * d) * id;
    o[6] = C * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
}
static void lpm_rgb_to_xyz(const float r[2], const float g[2], const float b[2], const float w[2], float out[9]) {
    float rz[3] = {r[0], r[1], 1.0f - r[0] + r[1]}, gz[3] = {g[0], g[1], 1.0f - g[0] + g[1]}, bz[3] = {b[0], b[1], 1.0f - b[0] + b[1]};
    float rgb3[9] = {rz[0], gz[0], bz[0], rz[1], gz[1], bz[1], rz[2], gz[2], bz[2]}; /* columns r g b */
    float rw = 1.0f / w[1];
    float w3[3] = {w[0] * rw, w[1] * rw, (1.0f - w[0] + w[1]) * rw};
    float inv[9]; mat3_inverse(rgb3, inv);
    float sc[3];
    for (int k = 0; k < 3; k++) sc[k] = inv[3 * k] * w3[0] + inv[3 * k + 1] * w3[1] + inv[3 * k + 2] * w3[2];
    for (int rrow = 0; rrow < 3; rrow++) for (int k = 0; k < 3; k++) out[3 * rrow + k] = rgb3[3 * rrow + k] * sc[k];
}
void orc_lpm_control_block(int shoulder, float soft_gap, float hdr_max, float exposure, float contrast, float shoulder_contrast,
                           const float saturation[3], const float crosstalk[3], uint32_t ctl[96]) { /* vk_tonemap.rs:122-325, LPM_CONFIG/COLORS_709_709 */
    (void)shoulder; (void)soft_gap;
    memset(ctl, 0, 96 * 4);
    contrast += 1.0f;
    float sat[3] = {saturation[0] + contrast, saturation[1] + contrast, saturation[2] + contrast};
    float mid_in = hdr_max * 0.18f * exp2f(-exposure), mid_out = 0.18f;
    float cs = contrast * shoulder_contrast;
    float z0 = -powf(mid_in, contrast), z1 = powf(hdr_max, cs) * powf(mid_in, contrast), z2 = powf(hdr_max, contrast) * powf(mid_in, cs) * mid_out;
    float z3 = powf(hdr_max, cs) * mid_out, z4 = powf(mid_in, cs) * mid_out;
    float tsb0 = -((z0 + (mid_out * (z1 - z2)) * (1.0f / (z3 - z4))) * (1.0f / z4));
    float tsb1 = (z1 - z2) * (1.0f / (z3 - z4));
    const float R[2] = {0.64f, 0.33f}, G[2] = {0.30f, 0.60f}, B[2] = {0.15f, 0.06f}, W[2] = {0.3127f, 0.3290f};
    float m[9]; lpm_rgb_to_xyz(R, G, B, W, m);
    float rs = 1.0f / (m[3] + m[4] + m[5]);
    float lumaW[3] = {m[3] * rs, m[4] * rs, m[5] * rs};
    float lumaT[3] = {m[3], m[4], m[5]};
    float rt = 1.0f / (lumaT[0] + lumaT[1] + lumaT[2]);
    for (int k = 0; k < 3; k++) lumaT[k] *= rt;
    float f[40]; memset(f, 0, sizeof f);
    f[0] = sat[0]; f[1] = sat[1]; f[2] = sat[2]; f[3] = contrast;
    f[4] = tsb0; f[5] = tsb1; f[6] = lumaT[0]; f[7] = lumaT[1];
    f[8] = lumaT[2]; f[9] = crosstalk[0]; f[10] = crosstalk[1]; f[11] = crosstalk[2];
    f[12] = 1.0f / lumaT[0]; f[13] = 1.0f / lumaT[1]; f[14] = 1.0f / lumaT[2];
    f[24] = shoulder_contrast; f[25] = lumaW[0]; f[26] = lumaW[1]; f[27] = lumaW[2];
    memcpy(ctl, f, 40 * 4);   /* ctl[0..9]; the fp16 half of the block (ctl[16..20]) is not used by the 32-bit LpmFilter */
}
static void lpm_filter_709(float *cr, float *cg, float *cb, const uint32_t ctl[96]) { /* LpmMap, ffx_lpm.h:727-832, all path flags false */
    float f[40]; memcpy(f, ctl, 40 * 4);
    float satR = f[0], satG = f[1], satB = f[2], contrast = f[3], tsbx = f[4], tsby = f[5], lT0 = f[6], lT1 = f[7], lT2 = f[8];
    float ctR = f[9], ctG = f[10], ctB = f[11], rl0 = f[12], rl1 = f[13], rl2 = f[14];
    float R = *cr, G = *cg, B = *cb;
    float rcpMax = 1.0f / fmaxf(fmaxf(R, G), B);
    float ratioR = powf(R * rcpMax, satR), ratioG = powf(G * rcpMax, satG), ratioB = powf(B * rcpMax, satB);
    float luma = G * lT1 + (R * lT0 + (B * lT2));
    luma = powf(luma, contrast);
    luma = luma * (1.0f / (luma * tsbx + tsby));
    float lumaRatio = ratioR * lT0 + ratioG * lT1 + ratioB * lT2;
    float ratioScale = clampf(luma * (1.0f / lumaRatio), 0.0f, 1.0f);
    R = clampf(ratioR * ratioScale, 0.0f, 1.0f); G = clampf(ratioG * ratioScale, 0.0f, 1.0f); B = clampf(ratioB * ratioScale, 0.0f, 1.0f);
    float capR = -ctR * R + ctR, capG = -ctG * G + ctG, capB = -ctB * B + ctB;
    float lumaAdd = clampf((-B) * lT2 + ((-R) * lT0 + ((-G) * lT1 + luma)), 0.0f, 1.0f);
    float t = lumaAdd * (1.0f / (capG * lT1 + (capR * lT0 + (capB * lT2))));
    R = clampf(t * capR + R, 0.0f, 1.0f); G = clampf(t * capG + G, 0.0f, 1.0f); B = clampf(t * capB + B, 0.0f, 1.0f);
    lumaAdd = clampf((-B) * lT2 + ((-R) * lT0 + ((-G) * lT1 + luma)), 0.0f, 1.0f);
    *cr = clampf(lumaAdd * rl0 + R, 0.0f, 1.0f); *cg = clampf(lumaAdd * rl1 + G, 0.0f, 1.0f); *cb = clampf(lumaAdd * rl2 + B, 0.0f, 1.0f);
}
void orc_present(const float *color, const uint32_t *ao, uint32_t n, uint32_t *packed_color, uint8_t *bgra8) {
    uint32_t ctl[96];
    const float sat[3] = {0.0f, 0.0f, 0.0f}, ct[3] = {1.0f, 0.5f, 1.0f / 32.0f};
    orc_lpm_control_block(0, 0.0f, 256.0f, 8.0f, 0.25f, 1.0f, sat, ct, ctl); /* vk_tonemap.rs:417-426 */
    for (uint32_t i = 0; i < n; i++) {
        uint32_t pk = orc_pack_b10g11r11(color + 4 * (size_t)i);
        if (packed_color) packed_color[i] = pk;
        float c[3]; orc_unpack_b10g11r11(pk, c);                         /* tonemap.comp.glsl:32 reads the stored image */
        float a = (float)(ao ? ao[i] : 255u) / 255.0f;
        c[0] *= a; c[1] *= a; c[2] *= a;
        if (fmaxf(fmaxf(c[0], c[1]), c[2]) > 0.0f) lpm_filter_709(&c[0], &c[1], &c[2], ctl); else c[0] = c[1] = c[2] = 0.0f; /* 0/0 in LpmMap: black stays black */
        for (int k = 0; k < 3; k++) { float v = powf(c[k], 1.0f / 2.2f); c[k] = v; }
        uint8_t *o = bgra8 + 4 * (size_t)i;                               /* swapchain B8G8R8A8_UNORM (renderer.rs:191-199) */
        o[0] = (uint8_t)(clampf(c[2], 0.0f, 1.0f) * 255.0f + 0.5f); o[1] = (uint8_t)(clampf(c[1], 0.0f, 1.0f) * 255.0f + 0.5f);
        o[2] = (uint8_t)(clampf(c[0], 0.0f, 1.0f) * 255.0f + 0.5f); o[3] = 255;
    }
}
#endif /* !ORC_F64 */
