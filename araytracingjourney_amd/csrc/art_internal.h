// art_internal.h -- shared declarations of libart's translation units (product code, gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/art.h"

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
};

// 64-byte binary traversal node: both child boxes inline, so one fetch decides both children.
//   q0 = lo0.xyz hi0.x | q1 = hi0.yz lo1.xy | q2 = lo1.z hi1.xyz | q3 = child0 child1 - -
// child >= 0: internal node index; child < 0: ~position of the triangle in leaf (Morton) order.
struct alignas(16) DevNode { float4 q[4]; };
// 48-byte triangle in leaf order: v0.xyz|prim  v1.xyz|tri-in-prim  v2.xyz|gid
struct alignas(16) DevTri { float4 v[3]; };

struct CameraArg { float view[16], view_inv[16], proj[16], proj_inv[16], camera_pos[3]; };

constexpr int kTile = 32;            // shard tile edge (pixels)
constexpr int kTilePixels = kTile * kTile;
constexpr uint32_t kNoHit = 0xFFFFFFFFu;
constexpr int kMaxLights = 16;

// ---- launch wrappers (art_build.hip / art_trace.hip) ---------------------------------------------------------
struct BuildInputs {
    const DevPrim *prims; uint32_t n_prims; const uint32_t *prim_first_tri; // device
    uint32_t T; uint32_t morton_bits;
};
struct Lbvh {               // canonical binary LBVH, device arrays
    uint32_t *leaf_gid;     // [T]
    uint64_t *keys;         // [T]
    int32_t *child;         // [2*(T-1)]
    float *node_lo, *node_hi; // [(T-1)*3]
    float *leaf_lo, *leaf_hi; // [T*3]
    DevTri *tris;           // [T] leaf order
    DevNode *nodes;         // [max(T-1,1)]
    uint32_t *tri_prim;     // [T] gid -> primitive
};
hipError_t lbvh_build(const BuildInputs &in, Lbvh &out, hipStream_t s); // allocates out.*, frees temporaries
void lbvh_free(Lbvh &l);

struct FrameArgs {
    CameraArg cam;
    uint32_t W, H;
    const uint32_t *tile_list; uint32_t n_tiles_owned; uint32_t tiles_x; // owned 32x32 tiles
    uint32_t n_local;          // n_tiles_owned * 1024
    const DevNode *nodes; const DevTri *tris; const DevPrim *prims; const uint32_t *tex_pool;
    const ArtLight *lights; uint32_t n_lights;
    float4 *hits;              // [n_local] t,u,v,gid
    float4 *contrib;           // [n_lights][n_local]
    float4 *shadow_rays;       // [2 * n_lights * n_local] dense per (light, pixel): o.xyz,tmax (<=0: none) | d.xyz,-
    uint32_t *counters;        // kCounterWords, zeroed every frame
    float4 *color; float *depth; float4 *normal; // full frame, row-major
    float4 *color_tiles;       // compact [n_local] (sharded mode) or nullptr
    uint32_t *shadow_bits;     // debug, [n_local] or nullptr
};
void launch_primary(const FrameArgs &a, hipStream_t s);
void launch_shade(const FrameArgs &a, hipStream_t s);
void launch_shadow(const FrameArgs &a, hipStream_t s);
void launch_accumulate(const FrameArgs &a, hipStream_t s);
void launch_query_closest(const DevNode *nodes, const DevTri *tris, const float4 *rays, uint32_t n, float4 *hits, uint32_t *cursors, hipStream_t s);
void launch_query_any(const DevNode *nodes, const DevTri *tris, const float4 *rays, uint32_t n, uint32_t *hit, uint32_t *cursors, hipStream_t s);
constexpr uint32_t kCounterWords = 1024; // [0] shadow rays, [1] hit pixels, [64..] primary cursors, [64+256..] shadow cursors, [64+512..] query cursors
void launch_untile(const float4 *gathered, uint32_t shard_count, uint32_t padded_tiles, uint32_t W, uint32_t H, float4 *frame, hipStream_t s);

// shard tile ownership: 32x32 tile (tx,ty) belongs to shard (tx + 5*ty) % count -- a diagonal interleave, so that
// every shard gets a near-equal number of tiles from every screen region (load balance)
__host__ __device__ inline uint32_t tile_owner(uint32_t tx, uint32_t ty, uint32_t count) { return count <= 1 ? 0u : (tx + 5u * ty) % count; }

} // namespace art
