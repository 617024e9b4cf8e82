/*
 * art_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Scalar C restatement of the reference's ray-tracing hot path
 * (/root/reference/src/vk_renderer/shaders/rt_lightning_shadows/{raytrace.rgen,light,ray_payload}.glsl, brdfs.glsl,
 * vk_camera.rs, lights.rs) plus a definition of what the closed Vulkan driver does
 * behind traceRayEXT (BVH build + traversal), which the reference does not contain.
 *
 * PARITY UNPINNED: the reference holds no golden vector, known-answer test or
 * fixture for this path and cannot be built or run here
 * (Rust + Vulkan RT).  This oracle is therefore pinned only by analytic known-answer
 * tests, a brute-force cross-check and an independent numpy restatement of the
 * shading maths (tests/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (araytracingjourney_amd/) never links or calls it.
 */
#ifndef ART_ORACLE_H
#define ART_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcScene OrcScene;

/* 80-byte light record == LightShaderData, lights.rs:69-82 == Light, light.glsl:1-12 */
typedef struct {
    float pos[3];    uint32_t type;           /* 0 point 1 spot 2 directional 3 area (lights.rs:88-93) */
    float dir[3];    uint32_t casts_shadows;
    float color[3];  float falloff_distance;
    float area_pos2[3]; float penumbra_angle;
    float area_pos3[3]; float umbra_angle;
} OrcLight;

/* 268-byte camera block == Uniform, vk_camera.rs:9-16 (column-major mat4 x4 + vec3) */
typedef struct {
    float view[16], view_inv[16], proj[16], proj_inv[16];
    float camera_pos[3];
} OrcCamera;

typedef struct {
    uint64_t primary_rays, shadow_rays;
    uint64_t hit_pixels;
    uint64_t n_int_primary, n_tri_primary;   /* canonical-LBVH visit counters */
    uint64_t n_int_shadow,  n_tri_shadow;
    uint64_t nonfinite_pixels;
} OrcStats;

OrcScene *orc_scene_create(void);
void      orc_scene_destroy(OrcScene *);
/* verts: nv x 12 floats (pos3 uv2 normal3 tangent4); idx: n_idx indices of idx_bytes (2|4);
 * tex: 3 layers (albedo, ORM, normal) x th x tw x RGBA8; model3x4: row-major object->world */
int orc_scene_add_primitive(OrcScene *, const float *verts, uint32_t nv, const void *idx, uint32_t n_idx,
                            uint32_t idx_bytes, const uint8_t *tex, uint32_t tw, uint32_t th,
                            const float model3x4[12]);
/* morton_bits: 30 or 63.  Builds the canonical binary LBVH (one triangle per leaf). */
int orc_scene_build(OrcScene *, int morton_bits);
uint32_t orc_scene_num_tris(const OrcScene *);
/* copies out the LBVH: leaf_gid[T], keys[T] (morton), child[2*(T-1)] (>=0 internal, <0 = ~leaf position),
 * node boxes lo/hi [T-1][3], leaf boxes [T][3], world triangle vertices [T][9]. Any pointer may be NULL. */
void orc_scene_get_lbvh(const OrcScene *, uint32_t *leaf_gid, uint64_t *keys, int32_t *child,
                        float *node_lo, float *node_hi, float *leaf_lo, float *leaf_hi, float *tri_verts);

/* host maths */
void orc_camera_from_params(const float pos[3], const float dir[3], float aspect, float fovy,
                            float znear, float zfar, OrcCamera *out);          /* vk_camera.rs:104-126,182-193 */
void orc_light_point(const float pos[3], const float color[3], float falloff, int casts, OrcLight *out);
void orc_light_spot(const float pos[3], const float dir[3], const float color[3], float falloff,
                    float penumbra, float umbra, int casts, OrcLight *out);
void orc_light_directional(const float dir[3], const float color[3], int casts, OrcLight *out);
void orc_light_area(const float pos[3], const float pos2[3], const float pos3[3], int invert_normal,
                    const float color[3], float falloff, float penumbra, float umbra, int casts, OrcLight *out);

/* rays: n x 8 floats (o.xyz, tmin, d.xyz, tmax) */
void orc_gen_primary(const OrcCamera *, uint32_t w, uint32_t h, float *rays);
/* mode 0 = canonical LBVH traversal, 1 = brute force over all triangles.
 * out_hit: n x 4 floats (t,u,v, unused) ; out_id: n x 2 ints (primitive index or -1, triangle id in primitive) */
void orc_trace_closest(const OrcScene *, const float *rays, uint32_t n, int mode, float *out_tuv, int32_t *out_id,
                       uint64_t *n_int, uint64_t *n_tri);
void orc_trace_any(const OrcScene *, const float *rays, uint32_t n, int mode, uint8_t *out_hit,
                   uint64_t *n_int, uint64_t *n_tri);

/* full frame: color (w*h*4), depth (w*h), normal (w*h*4) in row-major pixel order; rows [y0,y1).
 * Optional debug outputs (may be NULL): hit_tuv (w*h*4), hit_id (w*h*2), shadow_bits (w*h, bit i = light i shadowed,
 * bit 16+i = shadow ray i traced). */
void orc_render(const OrcScene *, const OrcCamera *, const OrcLight *, uint32_t n_lights, uint32_t w, uint32_t h,
                uint32_t y0, uint32_t y1, float *color, float *depth, float *normal, float *hit_tuv, int32_t *hit_id,
                uint32_t *shadow_bits, OrcStats *stats, int n_threads);

/* Packet-level visit counts of the same frame on the canonical LBVH: the frame is cut into block_w x block_h pixel blocks (8 x 8: what
 * one GPU wave traces as a packet); the primary rays of a block are one packet, its shadow rays towards light i another.  A node or
 * triangle a packet's rays touch (the union of their per-ray paths, each ray walked exactly as orc_render walks it) counts ONCE per
 * packet.  out = { nodes of primary packets, triangles of primary packets, nodes of shadow packets, triangles of shadow packets }.
 * This is the algorithmic fetch count of a packet tracer (bench.py prices its roofline with it); the per-ray counters of OrcStats
 * are the contract's figure, which charges every ray for every node.  stats: as orc_render's. */
void orc_packet_stats(const OrcScene *, const OrcCamera *, const OrcLight *, uint32_t n_lights, uint32_t w, uint32_t h,
                      uint32_t block_w, uint32_t block_h, uint64_t out[4], OrcStats *stats, int n_threads);

/* ray-traced ambient occlusion with XeGTAO's I/O contract (vk_xe_gtao.rs:17-23, :261-272, :295-333; consumer
 * tonemap.comp.glsl:33-34): inputs = the frame's depth + view-space normal outputs, output = 0..255 per pixel
 * (uint(pow(visibility, 2.2) * 255 + 0.5), 255 where nothing was hit).  spp cosine-weighted rays of length `radius`
 * per hit pixel, directions from the Hilbert-R2 noise of main_pass.comp.hlsl:48-65 with index + 288 * sample. */
void orc_render_ao(const OrcScene *, const OrcCamera *, uint32_t w, uint32_t h, const float *depth, const float *normal,
                   uint32_t spp, float radius, uint32_t *out_ao, uint64_t *n_rays, uint64_t *n_int, uint64_t *n_tri, int n_threads);

/* output packing + presentation: what the reference stores and shows.
 * pack: colour and normal as B10G11R11_UFLOAT_PACK32 (renderer.rs:268, vk_rt_lightning_shadows.rs:152), depth as R16_SFLOAT (:142);
 * present: tonemap.comp.glsl:29-40 = colour(from the packed image) * ao/255 -> LpmFilter(LPM_CONFIG_709_709) with the control
 * block of vk_tonemap.rs:122-325,:417-426 -> pow(1/2.2) -> B8G8R8A8_UNORM. */
uint32_t orc_pack_b10g11r11(const float rgb[3]);
void     orc_unpack_b10g11r11(uint32_t v, float rgb[3]);
uint16_t orc_pack_f16(float f);
void orc_lpm_control_block(int shoulder, float soft_gap, float hdr_max, float exposure, float contrast, float shoulder_contrast,
                           const float saturation[3], const float crosstalk[3], uint32_t ctl[96]);
void orc_present(const float *color, const uint32_t *ao /* may be NULL: 255 */, uint32_t n_pixels, uint32_t *packed_color, uint8_t *bgra8);

/* single-point shading for known-answer tests: shades a given hit without tracing the primary ray */
void orc_brdf_terms(float NdotL, float NdotV, float NdotH, float LdotH, float nc_NdotV, float nc_NdotL, float alpha,
                    float out[4]); /* D, V_fast, pow5 Schlick weight, Burley_local_sss */
void orc_light_eval(const OrcLight *, const float p[3], float nn_L[3], float radiance[3]);

#ifdef __cplusplus
}
#endif
#endif
