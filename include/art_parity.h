/*
 * art_parity.h -- the parity, rehearsal and measurement surface of libart: NOT part of the boundary a host application binds (that is include/art.h).
 *
 * Everything here is exported by the same libart.so -- the tests, bench.py and tools/ reach it through the same C ABI as the product calls, so what they
 * compare bit for bit is the product's binary, not a test build -- and falls into three groups:
 *   1. the EQUIVALENT FORMS of the path (ArtTuning / art_set_tuning: staged and per-ray frames, host-built trees, wave-plan targets ...) and read-backs of
 *      intermediate results (hit records, shadow bits, the trees) that the parity tests compare with the oracle;
 *   2. building blocks that art_mgpu_* (include/art.h) is made of, kept callable for rehearsals of other exchange loops (tile-buffer rings, strided un-tiles,
 *      stream / event hand-offs);
 *   3. measurement aids (per-stage event sums, device timestamps, graph mode).
 * A maintainer of the reference binds include/art.h (bindings/art_sys.rs is generated from it alone).
 */
#ifndef ART_PARITY_H
#define ART_PARITY_H
#include "art.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- parity / debug surface (not part of the reference's API; used by tests through this C ABI) ---- */
/* Which of the EQUIVALENT forms of the path a context runs.  All-zero = what the product runs; the other values exist so that tests can
 * show that every form gives the same frame bit for bit (tests/test_gpu_parity.py) and so that sweeps need no rebuild.  libart reads
 * nothing from the environment: a host application's environment cannot change which kernels or trees it gets.
 * Synchronises; the scene has to be built again afterwards (art_scene_build). */
typedef struct ArtTuning {
    uint32_t frame_form;        /* 0 one fused launch per frame, packet walks (k_frame) | 2 four staged launches, every ray by itself (the persistent per-ray tracer).  (1 -- staged launches of
                                 * packet walks -- went in round 4 with the other forms that had lost: the packet's beam as node step, PLOC trees, 6- / 7-wave instances) */
    uint32_t tree_builder;      /* 0 the context's default (binned SAH on the device; ART_FLAG_FAST_BUILD: LBVH topology) | 1 binned SAH on the host threads */
    uint32_t packet_wide;       /* the fused frame's packet walks: 0 = default (the 128-byte 4-wide float nodes), 1 = 4-wide, 2 = the 64-byte binary nodes */
    uint32_t primary_walk, shadow_walk, ao_walk; /* frame_form 2 (and the AO rays of any frame): one ray type's per-ray walk: 0 default | 2 binary nodes | 4 quantised 4-wide nodes */
    uint32_t block_order;       /* launch order of the 256-pixel blocks: 0 XCD-aware macro-blocks of 2x2 tiles | 1 identity | n: macro-blocks of n x n tiles */
    uint32_t fixed_waves;       /* 1: no adaptive wave plan (like ART_FLAG_FIXED_WAVES) */
    uint32_t split_fixed_steps; /* wave plan: a fixed packet-step target instead of the adaptive one (0: adaptive) */
    uint32_t split_min_steps;   /* wave plan: lowest target (0 = 150) */
    float split_alpha;          /* wave plan: fraction of the launch's fair share a wave may take before its block is split (0 = 0.7) */
    uint32_t ao_entry_off;      /* 1: AO rays start at the root instead of their pixel's entry node */
    uint32_t trace_chunk, trace_refill, trace_blocks; /* persistent per-ray tracer of THIS context's launches: slots per cursor pop, idle lanes that trigger a refill, resident blocks; 0 = presets */
    uint32_t hw_queues;         /* hardware queues the HOST gave the process (GPU_MAX_HW_QUEUES; 0 = HIP's default of 4): the wave plan counts min(frames in flight, this) launches in flight */
    uint32_t log;               /* to stderr: 1 build phase times, 2 wave-plan decisions, 4 every wave-plan poll */
    uint32_t wide_builder;      /* the 4-wide collapse of the binary tree: 0 level by level on the device | 1 one host thread (the form the device one is tested against) */
    uint32_t as_versions;       /* moving models: versions of the acceleration structure a context cycles through, 1..24 (0 = twice the frames in flight, 4 at least): a refit may run while as_versions - 1 older frames are in flight; 1 = refit in place, nothing in flight */
    float refit_rebuild_ratio;  /* art_trace rebuilds instead of refitting once ArtStats.refit_cost_ratio exceeds this (0 = 2.0; negative: never) */
    uint32_t trace_leaf_batch;  /* persistent per-ray tracer: lanes that must be waiting for a triangle test before the wave runs one, 1..64 (0 = presets: 1, AO rays 8) */
    uint32_t plan_moving_interval; /* wave plan: frames between two looks at the waves while the camera / the lights change every frame (0 = 32) */
    uint32_t refit_streams;     /* moving models: streams of their own the refits run on, beside the frames of the ring slot they precede (0 = min(frames in flight, 4); 0xFFFFFFFF: none -- every refit on its frame's stream, in front of it) */
    uint32_t refit_fold_nodes;  /* moving models: trees of this many 4-wide nodes and more make the quantised records and the cost inside the refit's own workgroups, with a cached share of the cost per batch (0 = 400 000; 1 = every tree: the tests' way to that form on small scenes) */
} ArtTuning;
int32_t art_set_tuning(ArtContext *ctx, const ArtTuning *tuning);
/* per-pixel primary hit record, row-major: tuv[4*i] = t,u,v,0 ; ids[2*i] = primitive index (-1 miss), triangle id */
int32_t art_read_hits(ArtContext *ctx, float *tuv, int32_t *ids, size_t n_pixels);
/* per pixel: bit i = light i shadowed, bit 16+i = shadow ray for light i traced (i < 16) */
int32_t art_read_shadow_bits(ArtContext *ctx, uint32_t *bits, size_t n_pixels);
/* arbitrary ray queries on the built scene.  rays: n x 8 floats (o.xyz, tmin, d.xyz, tmax), host memory. */
int32_t art_query_closest(ArtContext *ctx, const float *rays, uint32_t n, float *tuv, int32_t *ids);
int32_t art_query_any(ArtContext *ctx, const float *rays, uint32_t n, uint8_t *hit);
/* the device-built binary LBVH, in the oracle's canonical form (any pointer may be NULL):
 * leaf_gid[T], keys[T], child[2*(T-1)], node_lo/hi[(T-1)*3], leaf_lo/hi[T*3] */
int32_t art_get_lbvh(ArtContext *ctx, uint32_t *leaf_gid, uint64_t *keys, int32_t *child, float *node_lo,
                     float *node_hi, float *leaf_lo, float *leaf_hi);
/* the topology and node boxes the walks use over those leaves (child[2*(T-1)], node_lo/hi[3*(T-1)]): the binned-SAH tree by
 * default, the canonical tree with ART_FLAG_FAST_BUILD.  Node 0 is the root; child >= 0: internal node, < 0: ~leaf position. */
int32_t art_get_traversal_tree(ArtContext *ctx, int32_t *child, float *node_lo, float *node_hi);
/* the 4-wide collapse of that tree as the walks read it (new functionality: the reference's acceleration structures are opaque, vk_blas_builder.rs:88-170):
 * n_nodes records of 64 B (8-bit quantised child boxes: the per-ray walks) and of 128 B (float child boxes, children sorted along one axis: the packet
 * walks); node 0 is the root, child >= 0: node index, < 0: ~leaf position, INT32_MIN: absent (bit i of the records' valid masks clear).  Either pointer may be NULL; *n_nodes is always set. */
int32_t art_get_wide_nodes(ArtContext *ctx, void *quantised, void *floats, size_t capacity_nodes, uint32_t *n_nodes);


/* ---- building blocks of art_mgpu_* (rehearsals of other exchange loops) ---- */
/* render ring slot `slot`'s compact tiles straight into a caller-owned device buffer (e.g. the tensor handed to the
 * gather); bytes must equal padded * 12 KiB (4 KiB with ART_FLAG_PACKED_TILES) * frames per launch; NULL unbinds */
int32_t art_bind_color_tiles(ArtContext *ctx, uint32_t slot, void *dev_ptr, size_t bytes);
/* two buffers per slot: the slot's frames write them alternately (even / odd trips round the ring), so a frame never waits for the
 * exchange that is still reading the slot's previous tiles -- only for the one of two trips ago.  art_device_color_tiles and
 * art_read_color_tiles refer to the buffer the latest frame wrote. */
int32_t art_bind_color_tiles_pair(ArtContext *ctx, uint32_t slot, void *dev_even, void *dev_odd, size_t bytes);
/* n (1..8) buffers per slot, written in turn: trip t round the frame ring writes bufs[t % n], so a frame waits only for the exchange of n
 * trips ago.  With two, the host was found waiting at every trip boundary (the whole trip before last must have been exchanged);
 * four leave the slack a jittery exchange needs (profiles/README.md r1n). */
int32_t art_bind_color_tiles_ring(ArtContext *ctx, uint32_t slot, void *const *bufs, uint32_t n, size_t bytes);
int32_t art_read_color_tiles(ArtContext *ctx, void *dst, size_t bytes); /* host copy of the same buffer (tests) */
/* the same with shard s's tiles at gathered + s * shard_stride_tiles tiles: several frames gathered by ONE collective leave each
 * rank's frames back to back, so consecutive shards of one frame are a whole block of frames apart */
int32_t art_untile_gathered_strided(ArtContext *ctx, const void *gathered_dev, uint32_t shard_count, uint32_t shard_stride_tiles, void *frame_dev, void *hip_stream);
/* n_frames frames in ONE launch (the exchange of several ring slots by one collective): frame z's tiles start z * padded tiles into
 * every shard's buffer, its image is written at frames_dev + z * width * height elements */
int32_t art_untile_gathered_frames(ArtContext *ctx, const void *gathered_dev, uint32_t shard_count, uint32_t shard_stride_tiles, uint32_t n_frames, void *frames_dev, void *hip_stream);

/* make an external stream wait (on the device) for the most recently traced frame */
int32_t art_stream_wait_frame(ArtContext *ctx, void *hip_stream);
/* make the NEXT art_trace wait (on the device) for an external hipEvent_t, e.g. "the gather that read this slot's tiles
 * three frames ago has finished" */
/* art_trace + art_stream_wait_frame in one call (the per-frame host path of a sharded run); *slot_used = the ring slot the frame took */
int32_t art_trace_for_stream(ArtContext *ctx, void *hip_stream, uint32_t *slot_used);
int32_t art_wait_external_event(ArtContext *ctx, void *hip_event);


/* ---- measurement aids ---- */
/* graph mode: art_trace replays one captured hipGraph per ring slot (memset + 4 launches) instead of issuing them one by
 * one -- for host-bound runs (small per-GPU frames).  The capture is redone after a camera / light / extent / scene change;
 * per-stage timings are not available in this mode (only the whole frame). */
int32_t art_set_graph_mode(ArtContext *ctx, int32_t on);
/* two device timestamps on the frame streams (a measurement aid, e.g. "how long did frames i..j take while the ring stayed full"):
 * art_timestamp_mark(ctx, 0|1) records mark 0 / 1 behind the most recently traced frame on its stream; art_timestamp_elapsed waits for mark 1
 * and returns the time between the two. */
int32_t art_timestamp_mark(ArtContext *ctx, uint32_t which);
int32_t art_timestamp_elapsed(ArtContext *ctx, float *ms);

/* device time per stage (HIP events on the context's stream) summed over the frames traced since the previous call
 * (at most the last 128): sums_ms = primary, shade, shadow, accumulate, whole frame */
int32_t art_collect_timings(ArtContext *ctx, float sums_ms[5], uint32_t *n_frames);
/* traces ONE frame with the step-counting instance of the fused frame, waits for it and returns what every wave of that launch did:
 * items[2*i] = 8x8 pixel block (local pixel id / 64), items[2*i+1] = mask over its sixteen 2x2 cells (0: an idle wave), steps[i] = packet steps
 * (nodes + triangles visited, all walks of the wave).  *n = waves of the launch; at most cap are copied.  (tools/step_hist.py) */
int32_t art_sample_wave_steps(ArtContext *ctx, uint32_t *items, uint32_t *steps, uint32_t cap, uint32_t *n);


/* groups of launches of an ArtMgpu whose exchange has not been submitted yet (their frames may still run) and launches of the group that is still open.
 * Exchanges are submitted lazily -- from later launches or the flush -- so a host that also runs a control plane of its own between the ranks (barriers,
 * broadcasts) must not enter it while either is non-zero: ranks would meet in different collectives (art_mgpu_flush first). */
int32_t art_mgpu_pending(ArtMgpu *mg, uint32_t *groups_queued, uint32_t *launches_open);

#ifdef __cplusplus
}
#endif
#endif /* ART_PARITY_H */
