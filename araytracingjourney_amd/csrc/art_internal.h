// art_internal.h -- shared declarations of libart's translation units (product code, gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <utility>
#include "../../include/art.h"
#include "../../include/art_parity.h"

namespace art {

// ---- device-side scene tables -------------------------------------------------------------------------------
// PrimitiveInfo of raytrace.rgen.glsl:20-26 / vk_rt_descriptor_set.rs:31-38, extended with what the Vulkan
// runtime supplies implicitly (sampler dimensions, gl_ObjectToWorldEXT / gl_WorldToObjectEXT of the instance).
struct DevPrim {
    const float *vertices;      // 48-byte interleaved vertices of this primitive
    const void *indices;        // u16 or u32 triples
    uint32_t texture_offset;    // first texel (u32 RGBA8) of the 3-layer array in the texture pool
    uint32_t single_index_size; // 2 | 4
    uint32_t tw, th;
    uint32_t first_tri, n_tri;
    float o2w[12];              // row-major 3x4
    float w2o[12];
    uint32_t masked, pad_;      // a primitive that left the structure without a build (art_scene_set_primitive_enabled): its triangles are written "nowhere" by the next refit
};
// "Nowhere": the point box at 3e38 that absent children of a 4-wide node have carried since round 1 -- no ray passes it in any form of the slab test (|t| >= 3e38 on
// every axis).  A masked triangle's record is one, and a union skips such boxes (a node with nothing else below it is nowhere itself).
constexpr float kNowhere = 3.0e38f;
__host__ __device__ inline bool box_nowhere(float lo_x) { return lo_x >= kNowhere; }

// 64-byte binary traversal node: both child boxes inline, so one fetch decides both children.
//   q0 = lo0.xyz hi0.x | q1 = hi0.yz lo1.xy | q2 = lo1.z hi1.xyz | q3 = child0 child1 - -
// child >= 0: internal node index; child < 0: ~position of the triangle in leaf (Morton) order.
struct alignas(16) DevNode { float4 q[4]; };
// 64-byte 4-wide node with 8-bit quantised child boxes (the structure the tracer walks by default):
//   origin.xyz | exps = ex | ey<<8 | ez<<16 | valid_mask<<24   (scale_k = 2^(e_k-127), bit pattern e_k<<23)
//   q[0..2] = lo x/y/z, q[3..5] = hi x/y/z, byte c of each word = child c;  child box = origin + q*scale (exact fma),
//   rounded outwards at build time so that it contains the child's float box exactly
//   child[c] >= 0: wide node index; < 0: ~position of a triangle in leaf order
struct alignas(16) DevNode4 { float ox, oy, oz; uint32_t exps; uint32_t q[6]; uint32_t spare[2]; int32_t child[4]; };
static_assert(sizeof(DevNode4) == 64, "DevNode4 layout");
// 128-byte 4-wide node with full-precision child boxes, for the packet walk: a wave's iteration is bound by the latency of one
// dependent (scalar) node fetch, so halving the number of iterations matters more than the bytes.
//   box[c] = lo.xyz hi.xyz of child c, children sorted by box centre along axis pad[0] (the packet walk's front-to-back order); an absent
//   child is a point box at +3e38, which no slab test passes
struct alignas(16) DevNodeW { float box[4][6]; int32_t child[4]; uint32_t valid; uint32_t pad[3]; };
static_assert(sizeof(DevNodeW) == 128, "DevNodeW layout");
// reference of an absent child in both 4-wide records: INT32_MIN, the packet walk's own "take the next node from the stack" value (no leaf sits at position 2^31 - 1)
constexpr int32_t kAbsentChild = (int32_t)0x80000000;
// 64-byte triangle in leaf order, everything accept() needs in the form it needs it (one 64-byte scalar load for a packet):
//   f[0..2] v0 | f[3..5] e1 = v1 - v0 | f[6..8] e2 = v2 - v0 | f[9..11] box lo | f[12..14] box hi | f[15] gid (bits)
// e1, e2 and the box are the very float operations the tests would otherwise repeat per ray (one subtraction each; min / max of the three vertices),
// done once by k_leaves: Moeller-Trumbore and the triangle's own slab see the same bits.  (primitive, triangle-in-primitive follow from gid on the host.)
struct alignas(16) DevTri { float f[16]; };
static_assert(sizeof(DevTri) == 64, "DevTri layout");
// 144-byte shading record in leaf order: everything raytrace.rgen.glsl:107-114 fetches through PrimitiveInfo -> indices ->
// three 48-byte vertices, gathered once at build time so that hit reconstruction is one dependent fetch instead of three.
//   f[0..8] object-space positions p0 p1 p2 | f[9..14] uv0 uv1 uv2 | f[15..23] normals | f[24..32] tangent.xyz | f[33] v0.tangent.w | f[34] primitive id (bits)
struct alignas(16) DevShadeTri { float f[36]; };

struct CameraArg { float view[16], view_inv[16], proj[16], proj_inv[16], camera_pos[3]; };

constexpr int kTile = 32;            // shard tile edge (pixels)
constexpr int kTilePixels = kTile * kTile;
constexpr uint32_t kNoHit = 0xFFFFFFFFu;
constexpr int kMaxLights = 16;        // light records that travel in the kernel arguments
constexpr uint32_t kMaxLightsTotal = 1024; // art_set_lights' bound (records 16.. live in a device table per ring slot)
constexpr uint32_t kMaxFrameSlots = 24; // ring slots of a context (more than ~22 streams in use stall the command processor)
constexpr uint32_t kTileRingMax = 8;    // caller-owned compact tile buffers per ring slot (art_bind_color_tiles_ring)
int32_t ring_rewind(ArtContext *ctx);   // art_api.hip: waits for every frame in flight, then the next art_trace is launch 0 again (ring slot 0, first tile buffer)
void set_last_error(const char *msg);   // art_api.hip: the thread's art_last_error() string, for entry points that live in other files

// ---- launch wrappers (art_build.hip / art_trace.hip) ---------------------------------------------------------
struct BuildInputs {
    const DevPrim *prims; uint32_t n_prims; const uint32_t *prim_first_tri; // device
    uint32_t T; uint32_t morton_bits;
};
// Scratch memory of the builders: ONE device allocation that a context keeps from build to build (it only grows), cut up by each build phase.  The phases used to
// hipMalloc / hipFree their three dozen temporaries one by one -- 40 us apiece with the GPU idle in between: 2.1 of the 6.8 ms a rebuild of config 2 took.
// A phase reserves what it needs (nothing of an earlier phase is live: every phase ends with a stream synchronisation), then takes its pieces.
struct Arena {
    char *base = nullptr; size_t cap = 0, off = 0;
    uint64_t generation = 0;   // bumped by every reserve(): a pointer taken in an earlier phase is stale (take_checked in sanitizer / debug builds)
    static size_t pad(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    hipError_t reserve(size_t bytes) {
        off = 0; generation++;
        if (bytes <= cap) return hipSuccess;
        if (base) (void)hipFree(base);
        base = nullptr; cap = 0;
        hipError_t e = hipMalloc(&base, bytes + bytes / 8);   // (some room: the next scene is rarely the same size to the byte)
        if (e == hipSuccess) cap = bytes + bytes / 8;
        return e;
    }
    template <class T> T *take(size_t n) { T *p = reinterpret_cast<T *>(base + off); off += pad(n * sizeof(T)); return p; }   // the caller reserved the sum of pad(...)
    void release() { if (base) (void)hipFree(base); base = nullptr; cap = 0; off = 0; }
};
struct Lbvh {               // canonical binary LBVH, device arrays
    Arena *arena = nullptr; // host: the context's scratch for the build phases (not owned; null: a phase allocates and frees its own)
    uint32_t *leaf_gid;     // [T]
    uint64_t *keys;         // [T]
    int32_t *child;         // [2*(T-1)]
    float *node_lo, *node_hi; // [(T-1)*3]
    float *leaf_lo, *leaf_hi; // [T*3]
    DevTri *tris;           // [T] leaf order
    DevNode *nodes;         // [max(T-1,1)]
    uint32_t *tri_prim;     // [T] gid -> primitive
    DevNode4 *wide;         // [n_wide] collapsed + quantised traversal structure
    DevNodeW *widef;        // [n_wide] the same topology with float boxes (packet walk)
    uint32_t n_wide;
    bool canon_boxes = false;          // host: node_lo / node_hi of the canonical tree hold its boxes (else: not computed yet, binary_refit does it on demand)
    uint32_t log = 0;                  // host: ArtTuning.log of the context that builds (bit 0: build phase times to stderr)
    std::vector<uint32_t> wide_levels; // host: first wide node of every level of the collapse (breadth-first numbering), then n_wide -- the refit goes through them bottom-up
    DevShadeTri *shade_tris; // [T] leaf order
    uint32_t *leaf_parent;  // [T] the 4-wide node that holds a leaf, [n_wide] the one that holds a node (root: ~0): made with the first refit (launch_wide_parents), else null
    uint32_t *node_parent;
    // the refit's work lists (refit_lists_build, made with the parents): the tree below its top levels is cut into batches of neighbouring subtrees, one workgroup each
    uint32_t *sub_nodes = nullptr, *sub_leaves = nullptr, *sub_off = nullptr;   // device: batch b's nodes (deepest level first) / leaves; batch nb = the crown above the batches; sub_off: see k_refit_sub
    uint32_t sub_batches = 0, sub_levels = 0;                                     // host: batches (without the crown); levels of the tree
    std::vector<uint32_t> batch_prim_off, batch_prim_ids;                         // host: the primitives whose triangles lie in batch b: ids [off[b], off[b + 1]) -- a refit launches the batches that hold a primitive that moved
    int32_t *trav_child;    // [2*(T-1)] topology of the traversal nodes when it is not the canonical one (sah_build), else null
    float *trav_lo, *trav_hi; // [(T-1)*3]
    char *block = nullptr;  // host: ONE allocation behind the arrays lbvh_build makes (leaf_gid .. shade_tris, cbounds) and room for the traversal tree's three (res_trav_*): two dozen
                            // hipMalloc per build were 0.6-1 ms inside build_ms; lbvh_free frees the block, never its pieces
    int32_t *res_trav_child = nullptr; float *res_trav_lo = nullptr, *res_trav_hi = nullptr;   // in the block, for whichever builder makes a traversal tree (trav_* stay null until one does)
    uint32_t *cbounds;      // [6] the bounds of the triangle-box centroids the Morton keys were made over, as order-preserving keys (lo xyz, hi xyz): the root domain of the SAH bins
};
// art_jpeg.hip: baseline JPEG -> RGB8 (channels 3) or R8 (1), row-major
bool decode_jpeg(const uint8_t *data, size_t n, std::vector<uint8_t> &pixels, uint32_t &width, uint32_t &height, int &channels, std::string &err);
// allocates out.*, frees temporaries.  node_boxes false: the canonical tree's node boxes and the 64-byte node records are left for later -- a PREFER_FAST_TRACE build makes its
// own tree over the leaves and never reads them (binary_refit fills them in when the parity surface asks: out.canon_boxes)
hipError_t lbvh_build(const BuildInputs &in, Lbvh &out, hipStream_t s, bool node_boxes = true);
// the 4-wide collapses (DevNode4, DevNodeW): built on first use -- only the per-ray shadow/AO walks and ART_PACKET_WIDE need them
hipError_t wide_build(Lbvh &l, uint32_t T, hipStream_t s, bool on_host = false); // level by level on the device, or the one-thread host loop (A/B)
// PREFER_FAST_TRACE: rebuilds l.nodes as a binned-SAH tree over the same leaves (host threads); frames are unchanged by construction
hipError_t sah_build(Lbvh &l, uint32_t T, hipStream_t s);
// the binned SAH of sah_build, level by level on the device (art_sahdev.hip)
hipError_t sah_build_device(Lbvh &l, uint32_t T, hipStream_t s);
void launch_emit_nodes(Lbvh &l, uint32_t T, hipStream_t s);
// an empty launch from the translation unit of each builder: the runtime loads a unit's code object at its first launch (10-20 ms a process used to pay inside its first
// art_scene_build); art_create pays it instead
void build_prewarm(hipStream_t s);   // art_build.hip (LBVH, 4-wide collapse, refit, rocPRIM's sort and scans)
void sah_prewarm(hipStream_t s);     // art_sahdev.hip
// refit after a model moved (art_build.hip): the triangle records of a version of the acceleration structure from the shading records and that version's
// primitive table; its 4-wide nodes bottom-up, level by level (l.wide_levels); the tree's surface-area cost (2 doubles: sum of child half-areas, root half-area);
// the binary trees and node records from a version's triangles (on demand, synchronises)
// prims_host / touched: PINNED host memory the kernels read in place (the version's staging copies: one byte per primitive -- whose triangles to make again); mark: one word
// per 4-wide node, all zero between refits; acc: 4 doubles of device scratch, zero between launches; result: 4 doubles, pinned host memory (cost sum, root half-area, start / end stamps)
struct RefitArgs {
    uint32_t T, n_wide, n_prims;
    const uint32_t *sub_nodes, *sub_leaves, *sub_off; uint32_t sub_batches, sub_levels;
    const DevShadeTri *shade; const DevPrim *prims_host; DevPrim *prims_dev; const uint8_t *touched;
    const uint32_t *leaf_parent, *node_parent; uint32_t *mark;
    DevTri *tris; DevNode4 *wide; DevNodeW *widef; double *acc, *result;
    const uint32_t *dirty; uint32_t n_dirty;   // the batches to run (device-visible list; null: all of them)
    double *batch_cost;                        // [sub_batches] this version's cost share of every batch (large trees: a batch that does not run keeps its share)
    bool fold;                                 // the quantised records and the cost in the refit's own workgroups (large trees: kFoldRequantNodes / ArtTuning.refit_fold_nodes)
};
constexpr uint32_t kFoldRequantNodes = 400000;   // trees of this many 4-wide nodes and more: the refit's workgroups make the quantised records and the cost themselves (art_build.hip launch_refit)
void launch_refit(const RefitArgs &r, hipStream_t s);
void launch_wide_parents(uint32_t n_wide, const DevNodeW *widef, uint32_t *leaf_parent, uint32_t *node_parent, hipStream_t s);
hipError_t refit_lists_build(Lbvh &l, uint32_t T, hipStream_t s);   // after launch_wide_parents (synchronises: a one-off of the scene's first version ring)
void launch_wide_cost(uint32_t n_wide, const DevNodeW *widef, DevNode4 *wide /*null: cost only*/, double *acc, double *result, hipStream_t s);
hipError_t binary_refit(Lbvh &l, uint32_t T, const DevTri *tris, hipStream_t s);
struct TraceTune { uint32_t chunk, refill, blocks, leaf_batch; }; // overrides of the persistent per-ray tracer's presets (ArtTuning.trace_chunk / trace_refill / trace_blocks; 0 = the preset): they travel with every launch
void lbvh_free(Lbvh &l);
// the traversal tree's arrays: the room lbvh_build reserved for them, else allocations of their own (lbvh_free tells which by the block)
hipError_t lbvh_claim_trav(Lbvh &l, uint32_t NI);

// float32 -> unsigned small float (5 exponent bits, MB mantissa bits), round to nearest even; negatives -> 0, overflow -> +Inf
template <int MB> __host__ __device__ inline uint32_t pack_ufloat(float f) {
    uint32_t u;
#ifdef __HIP_DEVICE_COMPILE__
    u = __float_as_uint(f);
#else
    std::memcpy(&u, &f, 4);
#endif
    uint32_t e8 = (u >> 23) & 255u, m = u & 0x7FFFFFu;
    if (e8 == 255u && m) return (31u << MB) | 1u;
    if (u >> 31) return 0;
    if (e8 == 255u) return 31u << MB;
    int e = (int)e8 - 127 + 15;
    if (e >= 31) return 31u << MB;
    int shift = 23 - MB;
    uint32_t full = m | (e8 ? 0x800000u : 0u);
    if (e <= 0) { shift += 1 - e; e = 0; if (shift > 31) return 0; } else full &= 0x7FFFFFu;
    uint32_t q = full >> shift, rem = full & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    uint32_t out = ((uint32_t)e << MB) + q;
    return out > (31u << MB) ? (31u << MB) : out;
}
template <int MB> __device__ inline float unpack_ufloat(uint32_t v) {
    uint32_t e = v >> MB, m = v & ((1u << MB) - 1u);
    if (e == 31u) return m ? __uint_as_float(0x7FC00000u) : __uint_as_float(0x7F800000u);
    if (e == 0) return ldexpf((float)m, -14 - MB);
    return ldexpf((float)(m | (1u << MB)), (int)e - 15 - MB);
}
__device__ inline uint32_t pack_b10g11r11(float r, float g, float b) { return pack_ufloat<6>(r) | (pack_ufloat<6>(g) << 11) | (pack_ufloat<5>(b) << 22); }

struct FrameArgs {
    CameraArg cam;
    uint32_t W, H;
    const uint32_t *tile_list; uint32_t n_tiles_owned; uint32_t tiles_x; // owned 32x32 tiles
    uint32_t n_local;          // n_tiles_owned * 1024
    const uint32_t *block_order; // [n_local / 256] launch block -> 256-pixel block (XCD-aware order, art_api.hip setup_frame)
    const DevNode *nodes; const DevNode4 *wide; const DevNodeW *widef; const DevTri *tris; const DevShadeTri *shade_tris; const DevPrim *prims; const uint32_t *tex_pool;
    bool packet_wide;          // packet walks use the 128-byte 4-wide nodes (else the binary nodes)
    bool pipelined;            // several frames in flight (throughput-tuned launch) or one (latency-tuned)
    int trace_kind[3];         // how primary / shadow / AO rays are traced: 8 packet walk (the fused frame only), per-ray: 2 binary nodes, 4 quantised 4-wide nodes
    TraceTune tune;            // host: the context's overrides of the persistent tracer's presets
    // the light records travel BY VALUE with every launch, like the camera block: a frame in flight can never see a later art_set_lights
    // (a device-side table, however it is double-buffered, is overwritten while launches queued 16 frames ago still hold its address)
    ArtLight lights[kMaxLights]; uint32_t n_lights;
    const ArtLight *lights_more;   // lights kMaxLights .. n_lights - 1: a table of the frame's ring slot (device), uploaded on the slot's stream when the list changed; null up to kMaxLights
    uint32_t *pix_more;            // fused frame, more than 16 lights: [n_local] shadow rays traced for lights 16.. per pixel (pix_bits has 16 + 16 bits); else null
    float4 *hits;              // [n_local] t,u,v,gid
    float4 *contrib;           // [n_lights][n_local]
    float4 *shadow_rays;       // [2 * n_lights * n_local] dense per (light, pixel): o.xyz,tmax (<=0: none) | d.xyz,-
    uint32_t *counters;        // kCounterWords, zeroed every frame
    float4 *color; float *depth; float4 *normal; // full frame, row-major
    float4 *color_tiles;       // compact tile buffer (sharded mode) or nullptr: [n_local] RGB32F texels, 12 B each (the colour's alpha is the constant 1) ...
    bool tiles_packed;         // ... or B10G11R11 words (4 B per pixel, the reference's output image format)
    uint32_t *shadow_bits;     // debug, [n_local] or nullptr
    uint32_t *pix_bits;        // fused frame: [n_local] shadowed / traced bits per pixel (always written)
    bool keep_hits;            // fused frame: also store the hit records (art_read_hits)
    // fused frame: what each wave of the launch traces.  x = 8x8 pixel block (local pixel id / 64), y = which of its sixteen 2x2 cells
    // (bit = (y/2)*4 + x/2).  A block whose packet crawls (dense distant geometry: up to 0.5 ms for one wave) is dealt to 4 or 16 waves.
    const uint2 *wave_items; uint32_t n_wave_items;
    const uint32_t *tile_xy;   // [n_tiles_owned] x | y << 16 of each owned tile (tile units)
    uint32_t *wave_cost;       // [n_wave_items] packet steps each wave made, or nullptr
    // fused frame, several frames per launch (art_set_frames_per_launch): grid.y = frame b of the launch.  Frame b uses cam (b = 0) or
    // cam_more[b - 1], and writes its outputs b * (W * H) pixels (color, depth, normal), b * n_local (pix_bits, hits) and
    // b * tiles_stride (color_tiles) further on.  A launch costs ~7 us of machine time whatever it traces (profiles/README.md r1o).
    uint32_t batch; uint32_t tiles_stride; CameraArg cam_more[3];   // tiles_stride: TEXELS between two frames' tiles
};
constexpr uint32_t kMaxBatch = 4;
void launch_primary(const FrameArgs &a, hipStream_t s);
void launch_shade(const FrameArgs &a, hipStream_t s);
void launch_shadow(const FrameArgs &a, hipStream_t s);
void launch_accumulate(const FrameArgs &a, hipStream_t s);
bool launch_frame(const FrameArgs &a, hipStream_t s);      // the fused frame: primary + shade + shadow + accumulate in one launch; true: it also wrote a.wave_cost (a.wave_cost set and a counting instance exists)
void launch_frame_stats(const FrameArgs &a, uint32_t *out, hipStream_t s); // out[0] += shadow rays, out[1] += hit pixels
// The wave plan of the fused frame, made ON THE DEVICE behind a sampled frame (k_plan, art_trace.hip): from the steps every wave of that launch counted, every 8x8 block's level
// (0 one wave | 1 four quadrant waves | 2 sixteen cell waves) and, if a level changed that matters, the next table of wave items.  result (pinned host memory, read once the
// event behind the launch has fired): [0] items of the new table, [1] 1 = a new table was written, [2] blocks in four, [3] blocks in sixteen, [4] the step target, [5] slowest wave
struct PlanArgs {
    const uint2 *items_in; uint32_t n_items_in; const uint32_t *cost;   // the sampled launch: its table and what its waves counted
    uint8_t *level, *level_tmp; uint32_t n64;                             // per 8x8 block (device, persistent) + scratch
    uint32_t *worst;                                                      // [n64] scratch, zero between launches
    const uint32_t *order; uint32_t n256;                                 // launch order of the 256-pixel blocks (FrameArgs::block_order)
    uint2 *items_out; uint32_t cap;                                       // the table nobody reads at the moment
    float share; uint32_t min_steps, fixed_steps;                         // target = max(min_steps, share * steps of the sampled frame), or fixed_steps
    uint32_t *result;
};
void launch_plan(const PlanArgs &p, hipStream_t s);
// ambient occlusion on the frame's depth/normal outputs; occl: n_local*spp bytes; lut: spp+1 output values; cursors at counters[64+512..] are reused (queries never overlap a frame)
constexpr uint32_t kAoTableEntriesPerSample = 64 * 64;
void launch_ao_table(uint32_t spp, float4 *tab, hipStream_t s); // tab: spp * kAoTableEntriesPerSample float4
// pix: 2 * n_local float4 of scratch (per-pixel origin | start node, normal | noise index); tab: launch_ao_table's; entry_search: start the rays below the root
void launch_ao(const FrameArgs &f, uint32_t spp, float radius, uint8_t *occl, float4 *pix, const float4 *tab, bool entry_search, uint32_t *ao, const uint32_t *lut, hipStream_t s);
struct BvhView { const DevNode *nodes; const DevNode4 *wide; const DevTri *tris; int kind; TraceTune tune; }; // kind: 2 | 4
void launch_query_closest(const BvhView &b, const float4 *rays, uint32_t n, float4 *hits, uint32_t *cursors, hipStream_t s);
void launch_query_any(const BvhView &b, const float4 *rays, uint32_t n, uint32_t *hit, uint32_t *cursors, hipStream_t s);
// per-frame counter block (zeroed every frame): [64..] primary cursors, [64+256..] shadow cursors, [64+512..] query cursors,
// then 64 hit-pixel slots and 64 shadow-ray slots, each on its own 128-byte line: one word would serialise ~11 ns per atomic
// (32 640 waves on one address cost the shading kernel 0.3 ms)
constexpr uint32_t kCounterWords = 8192;
constexpr uint32_t kHitSlots = 1024, kShadowSlots = 1024 + 64 * 32, kSlotStride = 32, kSlotCount = 64;
// output packing + LPM tonemap (art_present.hip)
void lpm_control_block(bool shoulder, float soft_gap, float hdr_max, float exposure, float contrast, float shoulder_contrast, const float saturation[3],
                       const float crosstalk[3], uint32_t ctl[96]);
void launch_present(uint32_t n, const float4 *color, const float4 *normal, const float *depth, const uint32_t *ao, const uint32_t ctl[96], uint32_t *pcolor,
                    uint32_t *pnormal, uint16_t *pdepth, uint32_t *bgra, hipStream_t s);
// tile_slot[tile] = owner << 24 | index among the owner's tiles; shard_stride = tiles between two shards' buffers
// n_frames frames in one launch: frame z reads frame_stride tiles further into every shard's buffer and writes frame + z * W * H
void launch_untile_packed(const uint32_t *gathered, const uint32_t *tile_slot, uint32_t shard_stride, uint32_t n_frames, uint32_t frame_stride, uint32_t W, uint32_t H, uint32_t *frame, hipStream_t s);
void launch_untile(const float4 *gathered, const uint32_t *tile_slot, uint32_t shard_stride, uint32_t n_frames, uint32_t frame_stride, uint32_t W, uint32_t H, float4 *frame, hipStream_t s);

// shard tile ownership: 32x32 tile (tx,ty) belongs to shard (tx + 5*ty) % count -- a diagonal interleave, so that
// every shard gets a near-equal number of tiles from every screen region (load balance)
// Which shard owns which 32x32 tile (row-major table).  The tiles are walked along a Morton curve in groups of `count` neighbours and
// every group hands its tiles to the shards in a fresh pseudo-random order: each shard holds one tile of every neighbourhood (its work
// follows the frame's cost everywhere) and no lattice can beat against the scene's regularities -- the diagonal interleave
// (tx + 5 ty) mod count left the slowest of 8 shards 18 % above the mean on config 2 (profiles/README.md r1k).  Shares differ by <= 1 tile.
// Root relief (ArtConfig.root_relief): shard 0 composites besides tracing, so in `relief / 256` of the groups its tile goes to one of the other
// shards instead (each of them in turn).  Every rank of a job must create its context with the same value.
inline std::vector<uint8_t> shard_owner_table(uint32_t tiles_x, uint32_t tiles_y, uint32_t count, uint32_t relief) {
    std::vector<uint8_t> owner((size_t)tiles_x * tiles_y, 0);
    if (count <= 1) return owner;
    uint32_t relieved = 0;
    auto spread = [](uint32_t v) { v &= 0xFFFFu; v = (v | (v << 8)) & 0x00FF00FFu; v = (v | (v << 4)) & 0x0F0F0F0Fu; v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u; return v; };
    std::vector<std::pair<uint32_t, uint32_t>> order; // (Morton key, tile)
    order.reserve(owner.size());
    for (uint32_t ty = 0; ty < tiles_y; ty++) for (uint32_t tx = 0; tx < tiles_x; tx++) order.push_back({spread(tx) | (spread(ty) << 1), ty * tiles_x + tx});
    std::sort(order.begin(), order.end());
    std::vector<uint32_t> perm(count);
    for (size_t g = 0; g * count < order.size(); g++) {
        uint32_t state = (uint32_t)g * 2654435761u + 0x9E3779B9u; // one small generator per group: the table is the same on every rank
        for (uint32_t i = 0; i < count; i++) perm[i] = i;
        for (uint32_t i = count - 1; i > 0; i--) { state = state * 1664525u + 1013904223u; uint32_t j = (state >> 8) % (i + 1); std::swap(perm[i], perm[j]); }
        const bool relieve = relief && ((((uint32_t)g * 2246822519u) >> 24) < relief); // a fixed pseudo-random subset of the groups
        for (uint32_t i = 0; i < count && g * count + i < order.size(); i++) {
            uint32_t o = perm[i];
            if (relieve && o == 0) o = 1 + (relieved++ % (count - 1));
            owner[order[g * count + i].second] = (uint8_t)o;
        }
    }
    return owner;
}

} // namespace art
