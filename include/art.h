/*
 * art.h -- C ABI of libart, the MI355X-native (gfx950, HIP) ray-tracing core that stands in for the Vulkan
 * ray-tracing path of EdoardoLuciani/ARayTracingJourney.
 *
 * The reference has no FFI of its own: the path sits behind Rust methods that record Vulkan commands.  Each entry
 * point below names the reference interface (path:line under /root/reference/src/vk_renderer/) it replaces; the
 * binding a maintainer of the reference would add (a Rust `extern "C"` block) is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns 0 on success or a negative ART_E_* code; art_last_error() gives a thread-local message;
 *    nothing throws or unwinds across the boundary (the reference panics instead: renderer.rs uses unwrap/expect);
 *  - a context is single-threaded, like the reference's Rc<RefCell<..>> objects (renderer.rs:122-126);
 *  - plain pointers and sizes only; device work is enqueued on one HIP stream per context and art_sync() is the
 *    fence (the reference's analogue is the per-frame VkFence, renderer.rs:451-466);
 *  - there is NO CPU fallback: every compute entry point fails with ART_E_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef ART_H
#define ART_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ART_OK 0
#define ART_E_INVALID (-1)    /* bad argument */
#define ART_E_STATE (-2)      /* call out of order (e.g. trace before scene build) */
#define ART_E_NO_DEVICE (-3)  /* no HIP device / kernels not loadable */
#define ART_E_HIP (-4)        /* HIP runtime error, see art_last_error() */
#define ART_E_NOMEM (-5)

/* 48-byte interleaved vertex: model_reader.rs:22-35, gltf_model_reader.rs:176-199, raytrace.rgen.glsl:39-50 */
typedef struct ArtVertex {
    float pos[3];
    float uv[2];
    float normal[3];
    float tangent[4];
} ArtVertex;

/* 80-byte light record: lights.rs:69-82 (LightShaderData) == light.glsl:1-12 (Light) */
typedef struct ArtLight {
    float pos[3];
    uint32_t type; /* 0 point, 1 spot, 2 directional, 3 area (lights.rs:88-93) */
    float dir[3];
    uint32_t casts_shadows;
    float color[3];
    float falloff_distance;
    float area_pos2[3];
    float penumbra_angle;
    float area_pos3[3];
    float umbra_angle;
} ArtLight;

/* 268-byte camera block: vk_camera.rs:9-16 (Uniform) == raytrace.rgen.glsl:29-35; column-major mat4 */
#pragma pack(push, 1)
typedef struct ArtCamera {
    float view[16];
    float view_inv[16];
    float proj[16];
    float proj_inv[16];
    float camera_pos[3];
} ArtCamera;
#pragma pack(pop)

typedef struct ArtConfig {
    int32_t device;       /* HIP device ordinal; -1 = current device */
    uint32_t width;       /* initial extent (renderer.rs:140 takes vk::Extent2D) */
    uint32_t height;
    uint32_t morton_bits; /* 30 or 63; 0 = default (63) */
    uint32_t shard_rank;  /* screen-tile sharding: this context renders the 32x32 tiles owned by shard_rank of */
    uint32_t shard_count; /*   shard_count (0 or 1 = whole frame) */
    uint32_t flags;       /* ART_FLAG_* */
    uint32_t frames_in_flight; /* 0|1 = one; up to 24 (more than ~22 streams stall the command processor): a ring of per-frame streams + buffers like the reference's FrameData
                                  ring (renderer.rs:135, :300-318); art_trace then returns while up to N-1 older frames run */
    uint32_t root_relief;      /* sharded contexts, 0..255 (default 0 = equal shares), the same on every rank of a job: shard 0 -- the rank that also receives and un-tiles
                                  every frame when frames are assembled on rank 0 -- gives up its tile in root_relief / 256 of the tile groups to the other shards in turn, so
                                  that its share + compositing takes as long as the others' shares (art_shard_layout takes the same value) */
} ArtConfig;

#define ART_FLAG_FAST_BUILD 2u /* traversal nodes keep the LBVH topology (PREFER_FAST_BUILD); default: binned-SAH rebuild = PREFER_FAST_TRACE, vk_model.rs:968 */
#define ART_FLAG_PACKED_TILES 4u /* sharded contexts: the compact tile buffer (the gather's payload) holds B10G11R11_UFLOAT_PACK32 words -- the reference's colour
                                   image format (renderer.rs:268) -- 4 B per pixel instead of RGB32F's 12; art_untile_gathered then assembles the packed colour image */
/* (flag bit 8 was ART_FLAG_DEVICE_TREE until round 4 -- a PLOC-built tree, superseded by the binned SAH on the device: ignored) */
#define ART_FLAG_FIXED_WAVES 16u /* every 8x8 pixel block of a frame is traced by one wave, always.  Default: adaptive -- now and then a frame counts the packet
                                  * steps of each of its waves, and blocks whose wave outlasts the launch's fair share of the GPU (a packet crossing dense distant
                                  * geometry) are dealt to 4 or 16 waves in the following frames.  The image does not depend on it. */
#define ART_FLAG_TILE_OUTPUT 32u /* write the compact tile buffer even when the frame is not sharded (shard_count <= 1: one shard owning every tile): a job of ONE rank
                                  * then runs the whole art_mgpu_* path -- gather from itself, un-tile -- which is how the RCCL transport is exercised on a one-GPU machine */
#define ART_FLAG_DYNAMIC_SCENE 64u /* the host will move models (art_scene_set_model_matrix) or switch them in and out (art_scene_set_primitive_enabled): art_scene_build also makes the
                                   * ring of structure versions a refit writes into (ArtStats.first_move_ms otherwise falls into the first frame after the first move) */
#define ART_FLAG_KEEP_DEBUG 1u /* keep per-pixel hit records / shadow bits readable (art_read_hits, art_read_shadow_bits: include/art_parity.h) */

typedef struct ArtStats {
    uint64_t primary_rays;      /* W*H of the pixels this context owns */
    uint64_t shadow_rays;       /* shadow rays actually traced: hit && casts_shadows && N.L > 0 (raytrace.rgen.glsl:165) */
    uint64_t hit_pixels;
    uint64_t ao_rays;
    uint32_t num_triangles;
    uint32_t num_primitives;
    uint32_t num_nodes;         /* nodes of the traversal structure */
    uint32_t frame_launches; /* kernel launches per frame: 1 = fused frame kernel, 4 = primary / shade / shadow / accumulate */
    float build_ms;             /* last art_scene_build, device time */
    float frame_ms;             /* last art_trace, device time (events on the context's stream) */
    float trace_primary_ms, shade_ms, trace_shadow_ms, accumulate_ms;
    float ao_ms;                /* last art_trace_ao (ray generation + any-hit + resolve) */
    uint32_t split_blocks;      /* fused frame: 8x8 pixel blocks the current wave plan deals to 4 or 16 waves instead of one (see ART_FLAG_FIXED_WAVES) */
    float refit_ms;             /* last refit after art_scene_set_model_matrix, device time (triangle records + every box above them) */
    float refit_cost_ratio;     /* surface-area cost of the refitted tree over the cost of the tree as built (the latest refit whose figure has arrived); 1 = as built */
    uint32_t refits, rebuilds;  /* refits since art_create; builds art_trace started by itself because refit_cost_ratio passed ArtTuning.refit_rebuild_ratio (include/art_parity.h; default 2) */
    float first_move_ms;        /* host time art_trace spent making the ring of structure versions at the first move of a built scene (0 with ART_FLAG_DYNAMIC_SCENE: art_scene_build made it, inside versions_ms) */
    float versions_ms;          /* host time of making that ring, wherever it was made */
} ArtStats;

typedef struct ArtContext ArtContext;

const char *art_last_error(void);
int32_t art_device_count(void);

/* VulkanTempleRayTracedRenderer::new (renderer.rs:140) / Drop */
int32_t art_create(const ArtConfig *cfg, ArtContext **out);
int32_t art_destroy(ArtContext *ctx);
/* use an externally owned hipStream_t (e.g. torch's current stream); NULL restores the context's own stream.
 * Only for one frame in flight: a ring owns its streams (art_stream_wait_frame / art_wait_external_event in include/art_parity.h hand frames over to other streams). */
int32_t art_set_stream(ArtContext *ctx, void *hip_stream);
/* Several frames per launch (1..4; default 1; the fused frame only): every art_trace then traces n frames with ONE launch -- frame b with
 * the camera cams[b] of art_set_camera_batch (art_set_camera sets all n alike) -- and a ring slot holds n frames: every per-slot output,
 * the compact tile buffer included (bind n x the single-frame size: frame b's tiles follow frame b - 1's), is n frames back to back, and
 * the unit of art_trace, ring slots, art_frames_in_flight and art_frames_done is a launch.  A launch costs ~7 us of machine time whatever
 * it traces, which a 1/8 share of a 1080p frame (21 us of tracing) feels: 27.6 -> 24.4 -> 22.5 us per frame at 1 / 2 / 4 frames per launch
 * (profiles/README.md r1o).  art_read_*, art_device_* and art_get_stats refer to frame art_set_read_frame selects (default 0) of the
 * latest launch; art_trace_ao and art_present are not available with n > 1.  Synchronises; bound tile buffers are unbound. */
int32_t art_set_frames_per_launch(ArtContext *ctx, uint32_t n);
int32_t art_set_camera_batch(ArtContext *ctx, const ArtCamera *cams, uint32_t n);
int32_t art_set_read_frame(ArtContext *ctx, uint32_t b);
/* frame ring: number of slots and the slot the NEXT art_trace will use */
int32_t art_frames_in_flight(ArtContext *ctx, uint32_t *frames, uint32_t *next_slot);
/* host-side, non-blocking: have frames [first, first + count) of this context (counted from 0 in art_trace order, at most 128 back) all
 * finished?  *traced (optional) receives the number of frames traced so far.  An exchange that is SUBMITTED only once its frames are done
 * needs no device-side wait per frame: on a GPU that keeps tracing, every hipStreamWaitEvent packet in the exchange stream took ~40 us to
 * retire, which capped a 1/7 share at 46 us per frame (profiles/README.md r1n). */
int32_t art_frames_done(ArtContext *ctx, uint64_t first, uint32_t count, int32_t *done, uint64_t *traced);
/* add_model (renderer.rs:346) -> VkModel::create_blas geometry contract (vk_model.rs:886-943): one call per glTF
 * primitive.  idx_bytes = 2|4 (vk_model.rs:142-150); rgba8 = 3 layers albedo/ORM/normal of tw x th texels
 * (model_reader.rs:14-19); model3x4 = row-major object->world (vk_model.rs:358-363).  Data are copied. */
int32_t art_scene_add_primitive(ArtContext *ctx, const ArtVertex *verts, uint32_t n_verts, const void *indices,
                                uint32_t n_indices, uint32_t idx_bytes, const uint8_t *rgba8, uint32_t tw, uint32_t th,
                                const float model3x4[12], uint32_t *out_primitive_id);
int32_t art_scene_clear(ArtContext *ctx);
/* residency (vk_model.rs:334-345, renderer.rs:637-651): only models in the Device state are instanced in the TLAS.  A disabled
 * primitive keeps its id and its host copy but is not traced.  A primitive that is part of the built structure leaves and re-enters it WITHOUT a build: the refit
 * in front of the next art_trace (see art_scene_set_model_matrix) writes its triangles nowhere / back and shrinks / grows every box above them -- frames are those
 * of a scene built without / with it, bit for bit; its device arrays stay until the next art_scene_build.  A primitive that was disabled when the scene was built
 * (never uploaded) needs art_scene_build to appear: art_scene_needs_build says which case the context is in. */
int32_t art_scene_set_primitive_enabled(ArtContext *ctx, uint32_t primitive_id, int32_t enabled);
/* 1: the next art_trace would fail with ART_E_STATE until art_scene_build has run (primitives added, or enabled that the last build did not contain); 0: it would not */
int32_t art_scene_needs_build(const ArtContext *ctx);
/* VkModel::set_model_matrix (vk_model.rs:461-466) -> get_transform_model_matrix (:358-363) -> the instance record of the per-frame TLAS
 * (VkTlasBuilder::recreate_tlas every frame, renderer.rs:637-651, vk_tlas_builder.rs:38-233): primitives first_primitive .. first_primitive + n_primitives - 1
 * (one model's, art_scene_add_glb returns the range) get a new row-major object->world 3x4.  On a built scene nothing is built again: the NEXT art_trace
 * (or query) first REFITS on the device -- the world-space triangle records and every node box above them, the topology kept -- on that frame's own stream,
 * into the next of a small ring of versions of the structure (4 by default; ArtTuning.as_versions in include/art_parity.h), so frames in flight keep the scene they were launched
 * with and nothing waits unless every version is still being read (the reference's per-frame fence, renderer.rs:451-466).  Frames are those of a fresh
 * build, bit for bit (hits are structure-independent).  When the refitted tree's surface-area cost passes ArtTuning.refit_rebuild_ratio (default 2) times
 * the built tree's, art_trace builds again instead (ArtStats.rebuilds).  Every rank of an art_mgpu job must make the same calls. */
int32_t art_scene_set_model_matrix(ArtContext *ctx, uint32_t first_primitive, uint32_t n_primitives, const float model3x4[12]);
/* VkBlasBuilder::build_blas_from_geometry (vk_blas_builder.rs:88-170) + VkTlasBuilder::recreate_tlas
 * (vk_tlas_builder.rs:38-233): device LBVH over the world-space triangle soup. */
int32_t art_scene_build(ArtContext *ctx);

/* VkCamera::update_host_buffer (vk_camera.rs:104-126): the 268-byte block; and its producer.  A block holding a NaN or an infinity is ART_E_INVALID, and so
 * are parameters that have no finite matrices (a direction along the up axis (0, -1, 0) or of zero length, fovy 0, znear == zfar): the reference would
 * render NaN.  Rays that still come out non-finite (a finite but singular block) are misses, as in the oracle. */
int32_t art_set_camera(ArtContext *ctx, const ArtCamera *cam);
int32_t art_camera_from_params(const float pos[3], const float dir[3], float aspect, float fovy, float znear,
                               float zfar, ArtCamera *out);

/* VkLights::update_host_and_device_buffer (vk_lights.rs:81-139): n 80-byte records -- at most 1024 (the reference's list is a Vec behind an SSBO of n x 80 bytes, vk_lights.rs:89-91:
 * no bound but memory; libart carries the first 16 in the kernel arguments of every launch and the rest in a table per ring slot, so a frame in flight never sees a later list);
 * and their producers (lights.rs:144-159, :228-243, :281-296, :383-403).  art_read_shadow_bits (include/art_parity.h) reports the first 16 lights; ArtStats.shadow_rays counts all. */
int32_t art_set_lights(ArtContext *ctx, const ArtLight *lights, uint32_t n);
int32_t art_light_point(const float pos[3], const float color[3], float falloff, int32_t casts_shadows, ArtLight *out);
int32_t art_light_spot(const float pos[3], const float dir[3], const float color[3], float falloff, float penumbra,
                       float umbra, int32_t casts_shadows, ArtLight *out);
int32_t art_light_directional(const float dir[3], const float color[3], int32_t casts_shadows, ArtLight *out);
int32_t art_light_area(const float pos[3], const float pos2[3], const float pos3[3], int32_t invert_normal,
                       const float color[3], float falloff, float penumbra, float umbra, int32_t casts_shadows,
                       ArtLight *out);

/* VkRTLightningShadows::resize (vk_rt_lightning_shadows.rs:125-159) */
int32_t art_resize(ArtContext *ctx, uint32_t width, uint32_t height);
/* VkRTLightningShadows::trace_rays (vk_rt_lightning_shadows.rs:185-278): raygen + closest hit + light loop +
 * shadow rays + G-buffer stores of raytrace.rgen.glsl:77-200, asynchronous on the context's stream */
int32_t art_trace(ArtContext *ctx);
/* ray-traced ambient occlusion replacing VkXeGtao::compute_ao (vk_xe_gtao.rs:416-642) with the same I/O contract:
 * inputs = the depth + view-space normal outputs of the frame just traced (vk_xe_gtao.rs:295-333), output = one value
 * 0..255 per pixel in an R32_UINT-like buffer, as tonemap.comp.glsl:33-34 consumes it.  spp (1..64) cosine-weighted
 * any-hit rays of length `radius` per hit pixel (reference radius: 0.2 * 1.457, vk_xe_gtao.rs:17-18,:261), Hilbert-R2
 * noise (main_pass.comp.hlsl:48-65), final power 2.2 (vk_xe_gtao.rs:22).  Enqueued behind art_trace on its stream. */
int32_t art_trace_ao(ArtContext *ctx, uint32_t spp, float radius);
int32_t art_read_ao(ArtContext *ctx, void *dst, size_t bytes); /* width*height uint32 */
/* the step after the path (tonemap_layer.present, renderer.rs:566-615 / vk_tonemap.rs:469-552): packs the latest frame's
 * outputs as the reference stores them (colour + normal B10G11R11_UFLOAT_PACK32: renderer.rs:268, vk_rt_lightning_shadows.rs:152;
 * depth R16_SFLOAT: :142) and tonemaps like tonemap.comp.glsl:29-40 -- packed colour * ao/255 (255 if art_trace_ao has not run for
 * this frame) -> LpmFilter(LPM_CONFIG_709_709, control block of vk_tonemap.rs:417-426) -> pow(1/2.2) -> B8G8R8A8_UNORM. */
int32_t art_present(ArtContext *ctx);
int32_t art_read_present(ArtContext *ctx, void *dst_bgra8, size_t bytes);                       /* width*height*4 */
int32_t art_read_packed(ArtContext *ctx, void *color_b10g11r11, void *normal_b10g11r11, void *depth_f16); /* any may be NULL */
/* LpmData::new (vk_tonemap.rs:54-325): the 24 x uvec4 control block, host only */
int32_t art_lpm_control_block(int32_t shoulder, float soft_gap, float hdr_max, float exposure, float contrast, float shoulder_contrast,
                              const float saturation[3], const float crosstalk[3], uint32_t ctl[96]);
/* the fence (renderer.rs:451-466) */
int32_t art_sync(ArtContext *ctx);

/* get_color_output_image / get_output_depth_image / get_output_normal_image (vk_rt_lightning_shadows.rs:161-183):
 * fp32 RGBA colour (the value passed to imageStore, before the reference's lossy image formats), fp32 depth,
 * fp32 RGBA normal; row-major full frame, of the most recently traced frame.  Host copies (synchronising) and raw device
 * pointers (with a frame ring: valid until frames_in_flight - 1 more frames have been traced). */
int32_t art_read_color(ArtContext *ctx, void *dst, size_t bytes);
int32_t art_read_depth(ArtContext *ctx, void *dst, size_t bytes);
int32_t art_read_normal(ArtContext *ctx, void *dst, size_t bytes);
int32_t art_device_color(ArtContext *ctx, void **dev_ptr, size_t *bytes);
int32_t art_device_depth(ArtContext *ctx, void **dev_ptr, size_t *bytes);
int32_t art_device_normal(ArtContext *ctx, void **dev_ptr, size_t *bytes);

/* screen-tile sharding (new functionality, BASELINE.json): compact per-shard colour tiles for the RCCL gather, and
 * the un-tile step run by the root on the gathered buffer.  tile = 32x32 px of RGB32F = 12 KiB each: the colour without its alpha, which is
 * the constant 1 of imageStore(vec4(rho, 1)) (raytrace.rgen.glsl:197) -- three quarters of the bytes on the links, nothing lost; the un-tile writes
 * the RGBA32F frame. */
int32_t art_shard_tile_count(ArtContext *ctx, uint32_t *owned, uint32_t *padded);
/* host-only (no device needed): the row-major ids of the tiles shard_rank owns for a width x height frame cut for shard_count shards with
 * ArtConfig.root_relief = root_relief, in the order they sit in its compact buffer; *owned = their number, *padded = the largest count
 * over all shards (the per-rank gather size).  tiles may be NULL to query the counts; cap = capacity of tiles. */
int32_t art_shard_layout(uint32_t width, uint32_t height, uint32_t shard_count, uint32_t shard_rank, uint32_t root_relief, uint32_t *tiles,
                         uint32_t cap, uint32_t *owned, uint32_t *padded);
int32_t art_device_color_tiles(ArtContext *ctx, void **dev_ptr, size_t *bytes);
/* frame_dev NULL = the context's colour buffer; hip_stream NULL = the latest frame's stream */
int32_t art_untile_gathered(ArtContext *ctx, const void *gathered_dev, uint32_t shard_count, void *frame_dev, void *hip_stream);
/* what a caller that sizes buffers around a context needs to know (the multi-GPU frame below does) */
typedef struct ArtLayout {
    uint32_t width, height;
    uint32_t frames_in_flight;   /* ring slots */
    uint32_t frames_per_launch;  /* frames one art_trace traces = frames per ring slot */
    uint32_t shard_rank, shard_count;
    uint32_t tiles_owned, tiles_padded; /* 32x32 tiles this shard renders; the largest count over all shards (the per-rank gather size) */
    uint32_t tile_bytes;         /* bytes of one tile in the compact tile buffer: 12288 (RGB32F) or 4096 (ART_FLAG_PACKED_TILES); 0: the context writes none */
    uint32_t reserved;
} ArtLayout;
int32_t art_get_layout(ArtContext *ctx, ArtLayout *out);
/* ---- the sharded frame as one surface (new functionality, BASELINE.json north_star: "frames shard by screen tile across the 8 GPUs of one node
 * with an RCCL gather of the HDR buffer over xGMI"; the reference renders on one queue of one device, renderer.rs:188) ----------------------------
 * One process per GPU.  Every process creates its context with the shard art_mgpu_shard gives it, loads the same scene, and creates an ArtMgpu
 * from the SAME 128-byte id (rank 0 makes it with art_mgpu_unique_id and passes it on by whatever channel the host has: a file, a socket,
 * MPI, torch.distributed).  Then every rank calls, frame after frame and in the same order,
 *     art_set_camera(ctx, ..)   (all ranks the same camera)        art_mgpu_trace(mg)
 * and the assembled frames are found behind art_mgpu_flush / art_mgpu_read_frame (on rank 0, or with ART_MGPU_ROOT_SPREAD frame f on rank f mod world).
 * Inside: the rank's share is traced into a ring of compact tile buffers (4 per ring slot, written in turn); the tiles of `launches_per_gather`
 * launches travel as ONE exchange -- an ncclGather (RCCL, rccl.h:745) to rank 0, or one ncclGroupStart .. ncclGroupEnd of ncclSend / ncclRecv that
 * takes every frame to its own root -- on a stream of their own, submitted by the host once it has SEEN the group's frames finish (art_frames_done:
 * no device-side wait in front of the collective or of the next frames -- such waits cost the frames in flight their L2 contents, profiles/README.md
 * r1n), and one launch un-tiles a root's frames of the group.  The payload is the context's tile format: RGB32F -- the
 * HDR buffer, 12 B per pixel (its alpha is the constant 1) -- by default, B10G11R11 words with ART_FLAG_PACKED_TILES.
 * RCCL is loaded at art_mgpu_create (dlopen of librccl.so.1: libart itself does not link it); ART_E_NO_DEVICE if it cannot be. */
typedef struct ArtMgpu ArtMgpu;
#define ART_MGPU_ID_BYTES 128
#define ART_MGPU_SHARED 0u     /* rank 0 traces a share AND receives / un-tiles every frame */
#define ART_MGPU_DEDICATED 1u  /* rank 0 only receives and un-tiles; ranks 1..world-1 trace 1/(world-1) each */
#define ART_MGPU_RCCL 0u
#define ART_MGPU_HOST_EXCHANGE 1u /* the collective is the caller's function (rehearsals on one GPU, where RCCL refuses two ranks per device; other fabrics) */
#define ART_MGPU_ROOT_RANK0 0u   /* every frame is assembled on rank 0: one ncclGather per group of launches */
#define ART_MGPU_ROOT_SPREAD 1u  /* frame f (counted over the job, from 0) is assembled on rank f mod world: per group one ncclGroupStart .. ncclGroupEnd of ncclSend /
                                    ncclRecv, i.e. every frame's gather at once.  xGMI is point to point, one link per pair of GPUs: a single root receives over its own
                                    links only (2 GPUs: a half frame of RGB32F tiles, 12.4 MB at 1080p, per frame over ONE link); spread roots use every link in both
                                    directions.  art_mgpu_device_frame / art_mgpu_read_frame then give, on every rank, the newest frame that fell to it. */
/* must leave, ordered before anything enqueued on hip_stream afterwards, every rank's `bytes` bytes (rank order) in recv_dev on rank `root`
 * (recv_dev is NULL elsewhere); send_dev is complete when it is called, and nothing enqueued earlier on hip_stream still reads recv_dev (the library has
 * waited for the un-tile of the previous group).  Returns 0 or an error the library passes on as ART_E_HIP -- after which the ArtMgpu only reports that
 * error (frames of the failed group are lost; destroy it). */
typedef int32_t (*ArtMgpuExchangeFn)(void *user, const void *send_dev, size_t bytes, void *recv_dev, uint32_t root, void *hip_stream);
typedef struct ArtMgpuConfig {
    uint32_t rank, world;        /* this process; processes (= GPUs) of the job */
    uint32_t compositor;         /* ART_MGPU_SHARED | ART_MGPU_DEDICATED */
    uint32_t launches_per_gather;/* ring slots per collective; 0 = all of the ring (one collective per trip); rounded down to a divisor of the ring */
    uint32_t tile_buffers;       /* compact tile buffers per ring slot, 1..8; 0 = 4 */
    uint32_t transport;          /* ART_MGPU_RCCL | ART_MGPU_HOST_EXCHANGE */
    ArtMgpuExchangeFn exchange;  /* ART_MGPU_HOST_EXCHANGE only */
    void *exchange_user;
    uint32_t roots;              /* ART_MGPU_ROOT_RANK0 | ART_MGPU_ROOT_SPREAD (ART_MGPU_SHARED only) */
    uint32_t reserved;
} ArtMgpuConfig;
/* host only: the shard (ArtConfig.shard_rank / shard_count) rank `rank` of `world` creates its context with */
int32_t art_mgpu_shard(uint32_t rank, uint32_t world, uint32_t compositor, uint32_t *shard_rank, uint32_t *shard_count);
int32_t art_mgpu_unique_id(uint8_t id[ART_MGPU_ID_BYTES]);
/* ctx: built scene, final extent, frames in flight and frames per launch (change none of them while the ArtMgpu lives).  Collective: every rank calls it. */
int32_t art_mgpu_create(ArtContext *ctx, const ArtMgpuConfig *cfg, const uint8_t id[ART_MGPU_ID_BYTES], ArtMgpu **out);
/* one launch of this rank's share (art_trace: frames_per_launch frames with the context's current camera[s]) + its part of the exchange. */
int32_t art_mgpu_trace(ArtMgpu *mg);
/* every frame traced so far has been gathered and (rank 0) un-tiled when it returns.  Collective. */
int32_t art_mgpu_flush(ArtMgpu *mg);
/* a root (rank 0; every rank with spread roots), after art_mgpu_flush: the most recently traced frame it assembled (width x height RGBA32F, or B10G11R11 words with packed tiles) */
int32_t art_mgpu_device_frame(ArtMgpu *mg, void **dev_ptr, size_t *bytes);
int32_t art_mgpu_read_frame(ArtMgpu *mg, void *dst, size_t bytes);
/* frames traced / gathers submitted so far, and how many launches travel per gather */
int32_t art_mgpu_counts(ArtMgpu *mg, uint64_t *launches_traced, uint64_t *gathers, uint32_t *launches_per_gather);
/* before art_destroy of its context (it unbinds the tile buffers it owns from the context's ring slots) */
int32_t art_mgpu_destroy(ArtMgpu *mg);

int32_t art_get_stats(ArtContext *ctx, ArtStats *out);
/* ---- GLB ingest: the step right before the path (model_reader/gltf_model_reader.rs), host only ------------------ */
typedef struct ArtGlb ArtGlb;
/* MeshAttributeType / TextureType bits (model_reader.rs:5-20) */
#define ART_ATTR_VERTICES 1u
#define ART_ATTR_TEX_COORDS 2u
#define ART_ATTR_NORMALS 4u
#define ART_ATTR_TANGENTS 8u
#define ART_ATTR_INDICES 16u
#define ART_TEX_ALBEDO 1u
#define ART_TEX_ORM 2u
#define ART_TEX_NORMAL 4u
#define ART_TEX_EMISSIVE 8u
/* PrimitiveCopyInfo (model_reader.rs:52-72); image_format: 0 R8, 1 R8G8, 2 R8G8B8, 3 R8G8B8A8, 4 B8G8R8, 5 B8G8R8A8, 6..9 the R16 family */
typedef struct ArtGlbCopyInfo {
    uint64_t mesh_buffer_offset, mesh_size;
    uint64_t indices_buffer_offset, indices_size;
    uint64_t image_buffer_offset, image_size;
    uint32_t single_mesh_element_size, single_index_size;
    uint32_t image_format, image_width, image_height, image_mip_levels, image_layers, reserved;
} ArtGlbCopyInfo;
const char *art_glb_last_error(void);
/* GltfModelReader::open (gltf_model_reader.rs:55-150): coerce_format 0 none, 1 R8G8B8A8, 2 B8G8R8A8 (what VkModel asks for), 3 B8G8R8 */
int32_t art_glb_open(const char *path, int32_t normalize_vectors, int32_t coerce_format, ArtGlb **out);
int32_t art_glb_close(ArtGlb *glb);
int32_t art_glb_primitive_count(ArtGlb *glb, uint32_t *n);
/* copy_model_data_to_ptr (:156-281): dst NULL = sizing pass; *total = bytes the copy needs */
int32_t art_glb_copy_model_data(ArtGlb *glb, uint32_t attr_mask, uint32_t tex_mask, void *dst, size_t cap, ArtGlbCopyInfo *infos,
                                uint32_t n_infos, size_t *total);
/* get_primitives_bounding_sphere (:283-399) */
int32_t art_glb_bounding_sphere(ArtGlb *glb, float center[3], float *radius);
/* permute_pixels (:542-573): map[s] = destination byte of source byte s, or -1 */
int32_t art_glb_permute_pixels(const uint8_t *src, size_t src_len, uint32_t src_texel, const int32_t *map, uint32_t map_len,
                               uint32_t dst_texel, uint8_t *dst, size_t dst_cap);
/* add_model (renderer.rs:346 -> vk_model.rs:494-528): every primitive of the GLB into the scene */
int32_t art_scene_add_glb(ArtContext *ctx, ArtGlb *glb, const float model3x4[12], uint32_t *first_primitive_id, uint32_t *n_primitives);

#ifdef __cplusplus
}
#endif
#endif /* ART_H */
