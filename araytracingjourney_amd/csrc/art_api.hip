// art_api.hip -- the C ABI of include/art.h: context, scene tables, frame orchestration, host maths.
// Product code; gfx950 only; there is no CPU fallback anywhere in this file.
#include "art_internal.h"
#include <mutex>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

using namespace art;

namespace {

thread_local std::string g_err;
int32_t fail(int32_t code, const std::string &msg) { g_err = msg; return code; }
int32_t hipfail(hipError_t e, const char *what) { return fail(ART_E_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hipfail(e_, #x); } while (0)

struct HostPrim {
    std::vector<ArtVertex> verts;
    std::vector<uint8_t> indices; // original width
    uint32_t n_indices, idx_bytes;
    std::vector<uint8_t> tex;
    uint32_t tw, th;
    float o2w[12], w2o[12];
    bool enabled = true; // instanced in the acceleration structure (the reference's Device state, vk_model.rs:334-345); else kept on the host only
};

template <class T> struct DevBuf {
    T *p = nullptr; size_t n = 0;
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) n = count ? count : 1;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

} // namespace

void art::set_last_error(const char *msg) { g_err = msg ? msg : ""; }

// one frame in flight: its own stream and per-frame buffers, like the reference's FrameData ring (renderer.rs:135, :300-318)
struct FrameSlot {
    hipStream_t own = nullptr;
    DevBuf<uint32_t> d_counters, d_shadow_bits;
    DevBuf<float4> d_hits, d_contrib, d_shadow_rays, d_color, d_normal, d_color_tiles;
    DevBuf<float> d_depth;
    DevBuf<uint8_t> d_occl; DevBuf<uint32_t> d_ao; bool ao_valid = false;
    DevBuf<float4> d_ao_pix;           // per local pixel: the AO rays' origin | start node, world normal | noise index (k_ao_pixels)
    DevBuf<uint32_t> d_wave_cost;      // fused frame: packet steps of each wave of the slot's last launch (feedback for the wave plan)
    // more than 16 lights: records 16.. in a table of the slot's own (frames in flight on other slots keep theirs), uploaded on the slot's stream when the list changed since the
    // slot's last upload; the pinned staging copy is rewritten only once the upload that read it has finished
    DevBuf<ArtLight> d_lights_more; ArtLight *h_lights_more = nullptr; size_t h_lights_cap = 0; hipEvent_t lights_ev = nullptr; bool lights_ev_set = false; uint64_t lights_epoch = 0;
    DevBuf<uint32_t> d_pix_more;       // fused frame, more than 16 lights: shadow rays of lights 16.. per pixel
    DevBuf<uint32_t> d_pcolor, d_pnormal, d_bgra; DevBuf<uint16_t> d_pdepth; bool presented = false; hipEvent_t ao_ev[2] = {nullptr, nullptr};
    hipEvent_t done_alias = nullptr;   // the latest frame's completion is this ring event (fused frames: one record less per frame) instead of `done`
    float4 *ext_tiles = nullptr; size_t ext_tiles_bytes = 0; // caller-owned gather source (art_bind_color_tiles)
    static constexpr uint32_t kTileRing = kTileRingMax;
    float4 *ext_ring[kTileRing] = {}; uint32_t ext_ring_n = 0; // caller-owned buffers the slot's frames write in turn, one per trip round the frame ring (art_bind_color_tiles_ring)
    float4 *tiles_of_last = nullptr;   // where the slot's most recent frame wrote its tiles
    float4 *tiles_for(uint64_t frame_no, uint32_t F) { float4 *t = ext_tiles ? (ext_ring_n > 1 ? ext_ring[(frame_no / F) % ext_ring_n] : ext_tiles) : d_color_tiles.p; tiles_of_last = t; return t; }
    float4 *last_tiles() const { return tiles_of_last ? tiles_of_last : (ext_tiles ? ext_tiles : d_color_tiles.p); }
    hipEvent_t done = nullptr;       // recorded after the slot's last frame
    hipGraphExec_t graph = nullptr;  // the frame's launch sequence captured once (graph mode); dropped whenever an input changes
    void *wait_event = nullptr;      // external event the slot's next frame must wait for (art_wait_external_event)
    uint32_t as_version = 0;         // which version of the acceleration structure the slot's latest frame read (art_trace_ao and the read-backs follow it)
    void release() {
        d_counters.release(); d_shadow_bits.release(); d_hits.release(); d_contrib.release(); d_shadow_rays.release();
        d_color.release(); d_normal.release(); d_color_tiles.release(); d_depth.release(); d_occl.release(); d_ao.release(); d_ao_pix.release(); d_wave_cost.release(); d_lights_more.release(); d_pix_more.release();
        if (h_lights_more) (void)hipHostFree(h_lights_more); h_lights_more = nullptr; h_lights_cap = 0;
        if (lights_ev) (void)hipEventDestroy(lights_ev); lights_ev = nullptr; lights_ev_set = false; d_pcolor.release(); d_pnormal.release(); d_bgra.release(); d_pdepth.release();
    }
};
constexpr uint32_t kMaxFrames = kMaxFrameSlots;

// Which wave of the fused frame's launch traces what.  A launch lasts as long as its slowest wave, and an 8x8 packet that crosses dense
// distant geometry walks the union of 64 unrelated paths: up to 0.5 ms where the rest of the launch is done after 0.1 ms.  Every wave
// reports its packet steps; blocks that took many are dealt to four waves (4x4 pixels each) or sixteen (2x2) from the next plan on, heaviest first.
// The image does not depend on the plan (closest / any hit are structure- and packet-independent), only the launch's tail does.
struct WavePlan {
    bool enabled = true;               // false (ART_FLAG_FIXED_WAVES, ArtTuning.fixed_waves): every 8x8 block is one wave, always
    // A block is split when its wave makes more packet steps (nodes + triangles visited, all its walks) than the launch's fair share of the
    // machine would take anyway: alpha * (steps of the whole launch) * (launches in flight) / (wave slots of the GPU), at least min_steps.
    // With 16 full frames in flight nothing is split (a straggler hides behind the other launches, and split waves cost more steps in
    // total); one frame at a time, or a 1/8 share of a frame, is where the tail is the launch.
    float alpha = 0.7f;                // ArtTuning.split_alpha (0.5 .. 1 measured alike on 1/8 shares)
    uint32_t min_steps = 150;          // ArtTuning.split_min_steps
    uint32_t fixed_steps = 0;          // ArtTuning.split_fixed_steps: a fixed target instead (tests, experiments)
    uint32_t in_flight = 1;            // min(frames in flight, hardware queues)
    std::vector<uint32_t> order;       // launch order of the 256-pixel blocks (setup_frame)
    // The plan is made ON THE DEVICE (k_plan, art_trace.hip) behind a sampled frame, on that frame's stream: every block's level lives there, the two tables alternate so that
    // frames in flight keep theirs, and the host learns "a new table of n items" from eight pinned words once the event behind the launch has fired.  (Rounds 1-3: counts up, one
    // host thread through 32 640 blocks, table down -- 0.3-0.9 ms inside an art_trace call now and then, ten frames' time; tools/camera_leg_probe.py, profiles/README.md round 4.)
    DevBuf<uint2> d_items[2]; uint32_t n_items[2] = {0, 0}; int cur = 0;
    DevBuf<uint8_t> d_level, d_level_tmp; DevBuf<uint32_t> d_worst;
    uint32_t *h_result = nullptr, *dh_result = nullptr;   // pinned: PlanArgs::result
    uint32_t split1 = 0, split2 = 0;   // blocks the current table deals to four / sixteen waves
    hipEvent_t retire[2][kMaxFrames] = {}; bool retire_set[2] = {false, false}; // recorded on every frame stream when table i was left: it may be rewritten once they have all fired
    uint32_t cap = 0;                  // items a table (and the cost buffers) hold
    hipEvent_t cost_ready = nullptr; bool pending = false; int pending_table = 0;   // behind the sampled frame's k_plan
    hipStream_t plan_stream = nullptr; // k_plan runs here, behind the sampled frame's completion event: one workgroup for ~0.1 ms -- on the frame's own stream the slot's next frame stood behind it (5-10 % of a 20-frame burst)
    uint64_t next_sample = 0; uint32_t interval = 1;
    uint64_t last_sample = 0;          // the frame that was sampled last
    bool moved_since_poll = false;     // the view or the lights changed since the last plan came back: a new table is no reason to look again at once (the next one would differ too)
    uint32_t replans = 0;
    void release() {
        d_items[0].release(); d_items[1].release(); d_level.release(); d_level_tmp.release(); d_worst.release();
        if (h_result) (void)hipHostFree(h_result); h_result = nullptr; dh_result = nullptr;
        if (cost_ready) (void)hipEventDestroy(cost_ready); cost_ready = nullptr;
        if (plan_stream) { (void)hipStreamSynchronize(plan_stream); (void)hipStreamDestroy(plan_stream); } plan_stream = nullptr;
        for (int i = 0; i < 2; i++) for (uint32_t k = 0; k < kMaxFrames; k++) if (retire[i][k]) { (void)hipEventDestroy(retire[i][k]); retire[i][k] = nullptr; }
    }
};

// One version of what a frame reads of the acceleration structure.  A static scene has none (the build's arrays are read directly); the first
// art_scene_set_model_matrix makes a small ring of them: a refit writes the NEXT version while frames in flight still read the older ones, like the
// reference's per-frame TLAS (one VkTlasBuilder per FrameData, renderer.rs:300-318, :637-651).  Version 0 is the build's own arrays.
constexpr uint32_t kMaxAsVersions = 24;
struct AsVersion {
    DevTri *tris = nullptr; DevNodeW *widef = nullptr; DevNode4 *wide = nullptr; DevPrim *prims = nullptr;
    bool owned = false;                  // version 0 aliases c->bvh.* and c->d_prims
    // Pinned host memory the refit's kernels read and write IN PLACE (no copies in front of or behind the launches): the primitive table as of this refit, a byte per
    // primitive (moved since this version was written), and the refit's result (cost sum, root half-area, start / end device stamps).  d*: the device's addresses of the same.
    DevPrim *h_prims = nullptr, *dh_prims = nullptr; uint8_t *h_touched = nullptr, *dh_touched = nullptr; double *h_result = nullptr, *dh_result = nullptr;
    uint32_t *mark = nullptr;            // per 4-wide node (art_build.hip k_retri): all zero between refits
    uint32_t *h_dirty = nullptr, *dh_dirty = nullptr;   // pinned: the batches this refit runs (those that hold a primitive that moved since the version was written)
    double *batch_cost = nullptr; bool cost_cached = false;   // device: every batch's share of this version's cost (large trees); valid once a refit has run all batches
    double *acc = nullptr;               // 4 doubles of device scratch of the refit's last launch: zero between refits
    uint64_t used[kMaxFrameSlots] = {};  // frame number + 1 of the newest launch on each ring slot that read this version (0: none)
    bool aux[kMaxFrameSlots] = {};       // art_trace_ao / art_present ran behind that frame on the slot's stream
    hipEvent_t ready = nullptr; bool ready_known = true; uint32_t ready_slot = 0; // the refit that wrote it: recorded on ring slot ready_slot's stream
    bool result_pending = false;         // h_result is that refit's once `ready` has fired
    uint64_t epoch = 0;                  // which refit wrote it (0: the build)
};

struct ArtContext {
    ArtConfig cfg{};
    int device = 0;
    hipStream_t ext_stream = nullptr; // art_set_stream (single frame in flight only)
    uint32_t F = 1, last = 0;         // frames in flight; slot of the most recently submitted frame
    FrameSlot slot[kMaxFrames];
    uint32_t W = 0, H = 0;
    std::vector<HostPrim> prims;
    bool built = false, have_camera = false, frame_ready = false;
    // which of the equivalent forms this context runs: the defaults are the product, the others are reachable through art_set_tuning only (nothing reads the environment)
    ArtTuning tuning{};
    bool fused = true;        // the frame is ONE launch of packet walks (k_frame); frame_form 2: four staged launches, every ray by itself
    int tree_builder = 3;     // with fast_trace: 3 = binned SAH on the device (art_sahdev.hip), 1 = the same on the host threads (art_sah.hip)
    bool fast_trace = true;   // rebuild the traversal tree with the binned SAH after the LBVH (ART_FLAG_FAST_BUILD: keep the Karras tree)
    bool packet_wide = true;  // packets walk the 128-byte 4-wide float nodes (half the dependent node fetches); false: the 64-byte binary nodes
    int kind_primary = 8, kind_shadow = 8, kind_ao = 4; // 8 = packet walk over the binary nodes (coherent rays: primary, shadow); per-ray walks (AO, queries): 2 binary, 4 wide quantised (measured: profiles/README.md)
    uint32_t macro = 2;       // XCD-aware launch order: macro-blocks of macro x macro tiles (0: identity)
    bool ao_entry = true;     // AO rays start at the per-pixel entry node (k_ao_entry)
    bool wide_on_host = false; // ArtTuning.wide_builder
    DevBuf<float4> d_ao_tab; uint32_t ao_tab_spp = 0; // art_trace_ao's sample table and the sample count it was made for
    // device scene
    DevBuf<float> d_verts; DevBuf<uint8_t> d_indices; DevBuf<uint32_t> d_tex; DevBuf<DevPrim> d_prims; DevBuf<uint32_t> d_first_tri;
    std::vector<uint32_t> h_first_tri; // first global triangle id of every primitive slot (ascending): gid -> (primitive, triangle) on the host
    Lbvh bvh{};
    Arena arena;              // the build phases' scratch, kept from build to build (art_internal.h)
    std::vector<uint8_t> uploaded;   // which primitives' vertices / indices / texels are on the device, at the offsets a build over exactly this set computes (empty: nothing): a
                              // build over the same set -- the rebuild behind the refit's cost rule, a change of tuning -- uploads only the primitive table
    uint32_t T = 0;
    // moving models (art_scene_set_model_matrix): versions of the structure, the primitive table as the next refit will upload it
    std::vector<AsVersion> as; uint32_t as_cur = 0; bool xform_dirty = false;
    void *as_block = nullptr, *as_pinned = nullptr;   // ONE device allocation and ONE pinned one behind all the versions (six + three per version before).  versions_ms is something else: the refit streams' creation (a
                                                      // high-priority hardware queue each: 4-10 ms apiece) and the refit's work lists (host, 9 ms for config 2) -- ArtTuning.log bit 0 prints the parts
    // Refits run on streams of their own, one per ring slot (up to four): the refit in front of frame n of slot k then overlaps frame n - F, which still runs on that slot's
    // stream, instead of queueing behind it -- the slot's chain is frame, frame, frame with the refits beside it, and the frame waits for its refit's event.  (On the frame's
    // own stream a slot's cycle was refit + frame: a model moving every frame cost the ring a third of its depth, profiles/README.md round 4.)
    hipStream_t refit_stream[4] = {nullptr, nullptr, nullptr, nullptr}; uint32_t n_refit_streams = 0;
    std::vector<DevPrim> h_dev_prims;          // host copy of d_prims (build order), matrices kept current
    std::vector<uint64_t> prim_moved;          // per primitive: the refit (as_epoch numbering) that first shows its latest move; 0: where the build put it
    int64_t masked_tris = 0;                   // triangles of primitives disabled since the build (still in the arrays, written "nowhere")
    uint64_t as_epoch = 0, binary_epoch = 0;   // refits so far; the refit the binary trees / node records reflect
    double as_cost0 = 0.0; float refit_cost_ratio = 1.0f; uint32_t refits = 0, rebuilds = 0; float last_refit_ms = 0.f, first_move_ms = 0.f, versions_ms = 0.f;
    ArtCamera camera{};
    uint32_t B = 1, read_b = 0;       // frames per launch of the fused frame (art_set_frames_per_launch); which of them the read / device-pointer calls refer to
    ArtCamera cam_more[kMaxBatch - 1] = {}; // cameras of frames 1.. of a launch (frame 0: camera)
    std::vector<ArtLight> lights; uint64_t lights_epoch = 1;   // (bumped by every art_set_lights that changes the list)
    // frame
    std::vector<uint32_t> tile_list; uint32_t tiles_x = 0, tiles_y = 0, padded_tiles = 0, n_local = 0;
    DevBuf<uint32_t> d_tile_list;
    DevBuf<uint32_t> d_tile_xy;     // owned tile -> x | y << 16 (fused frame: no division per pixel lookup)
    DevBuf<uint32_t> d_tile_slot;   // un-tile table: tile -> owner << 24 | index among the owner's tiles (every shard's layout, setup_frame)
    DevBuf<uint32_t> d_block_order; // launch block -> 256-pixel block of the frame: one L2 (XCD) per screen region (setup_frame)
    WavePlan plan;                  // fused frame: wave -> (8x8 block, cells)
    static constexpr int kRing = 128;          // per-frame stage events kept for art_collect_timings
    hipEvent_t ev[kRing][5] = {};
    bool ev_fused[kRing] = {};                 // the frame was one launch: only ev[0] and ev[4] were recorded
    uint64_t frame_no = 0, collected_upto = 0;
    hipEvent_t mark[2] = {nullptr, nullptr};   // art_timestamp_mark
    bool traced = false;
    bool force_sample = false; // art_sample_wave_steps: the next fused frame counts its waves' steps whatever the plan's cadence
    bool graph_mode = false; // replay a captured hipGraph per slot instead of 5 launches + 6 event records (host-bound multi-GPU runs)
    uint32_t ao_spp = 0;
    ArtStats stats{};
    bool tiles_packed() const { return (cfg.flags & ART_FLAG_PACKED_TILES) != 0; }
    bool tiled() const { return cfg.shard_count > 1 || (cfg.flags & ART_FLAG_TILE_OUTPUT) != 0; } // writes the compact tile buffer beside the frame
    size_t tile_px_bytes() const { return tiles_packed() ? 4 : 12; } // B10G11R11 words or RGB32F (the colour without its constant alpha) in the compact tile buffer
    hipStream_t stream_of(uint32_t k) const { return (ext_stream && F == 1) ? ext_stream : slot[k].own; }
    hipStream_t main_stream() const { return stream_of(0); }
};

// Frame streams are kept for the life of the process and handed from a destroyed context to the next one: streams created after
// others were destroyed share hardware queues badly (a context made after another had been destroyed ran 60 % slower, profiles r1k).
static std::mutex g_stream_mutex;
static std::vector<std::pair<int, hipStream_t>> g_free_streams; // (device, stream)
static hipError_t acquire_stream(int device, hipStream_t *out) {
    {
        std::lock_guard<std::mutex> lock(g_stream_mutex);
        for (size_t i = 0; i < g_free_streams.size(); i++)
            if (g_free_streams[i].first == device) { *out = g_free_streams[i].second; g_free_streams.erase(g_free_streams.begin() + (long)i); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
static void release_stream(int device, hipStream_t s) {
    (void)hipStreamSynchronize(s);
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    g_free_streams.push_back({device, s});
}

namespace {

void affine_inverse(const float *m, float *o) { // row-major 3x4
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

int32_t use_device(ArtContext *c) {
    HIPC(hipSetDevice(c->device));
    return ART_OK;
}

void drop_graphs(ArtContext *c) {
    for (uint32_t k = 0; k < kMaxFrames; k++)
        if (c->slot[k].graph) { (void)hipStreamSynchronize(c->stream_of(k)); (void)hipGraphExecDestroy(c->slot[k].graph); c->slot[k].graph = nullptr; } // never destroy a graph in flight
}

int32_t sync_all(ArtContext *c);
// the wide collapse is host work on the finished binary tree: done lazily, the first time a walk that needs it is launched
int32_t ensure_wide(ArtContext *c, bool needed) {
    if (!needed || c->bvh.wide) return ART_OK;
    int32_t r = sync_all(c); if (r) return r;
    hipEvent_t e0, e1; float ms = 0;
    HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
    HIPC(hipEventRecord(e0, c->main_stream()));
    hipError_t e = wide_build(c->bvh, c->T, c->main_stream(), c->wide_on_host);
    if (e == hipSuccess) e = hipEventRecord(e1, c->main_stream());
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess) return hipfail(e, "wide_build");
    c->stats.build_ms += ms;
    return ART_OK;
}

int32_t sync_all(ArtContext *c) {
    for (uint32_t i = 0; i < c->n_refit_streams; i++) HIPC(hipStreamSynchronize(c->refit_stream[i]));   // (a refit no frame has waited for yet)
    if (c->plan.plan_stream && c->plan.pending) { for (uint32_t k = 0; k < c->F; k++) HIPC(hipStreamSynchronize(c->stream_of(k))); HIPC(hipStreamSynchronize(c->plan.plan_stream)); }   // (a plan behind a sampled frame)
    for (uint32_t k = 0; k < c->F; k++) HIPC(hipStreamSynchronize(c->stream_of(k)));
    return ART_OK;
}

// ---- versions of the acceleration structure (moving models) ------------------------------------------------------------------------------
struct AsPtrs { const DevTri *tris; const DevNodeW *widef; const DevNode4 *wide; const DevPrim *prims; };
AsPtrs as_ptrs(const ArtContext *c, uint32_t v) {
    if (c->as.empty()) return AsPtrs{c->bvh.tris, c->bvh.widef, c->bvh.wide, c->d_prims.p};
    const AsVersion &V = c->as[v];
    return AsPtrs{V.tris, V.widef, V.wide, V.prims};
}
uint64_t as_epoch_of(const ArtContext *c, uint32_t v) { return c->as.empty() ? 0 : c->as[v].epoch; }
// everything that reads them has finished (the caller synchronised)
void as_release(ArtContext *c) {
    for (AsVersion &V : c->as) {
        if (V.ready) (void)hipEventDestroy(V.ready);
    }
    (void)hipFree(c->as_block); c->as_block = nullptr;                         // every version's device arrays
    if (c->as_pinned) (void)hipHostFree(c->as_pinned); c->as_pinned = nullptr; // every version's staging memory
    c->as.clear(); c->as_cur = 0;
}
// the first move of a built scene: the ring of versions (ArtTuning.as_versions; default 4: one more than the reference's frames in flight, renderer.rs:135 -- measured on
// config 2 with a model of 164 k triangles moving every frame, 16 ring slots: 0.70 / 0.38 / 0.28 / 0.25 / 0.26 ms per frame with 1 / 2 / 3 / 4 / 8 versions), every one a copy of
// the build's arrays -- the topology (child references, valid masks, sort axes) is never written again -- and the cost of the tree as built
int32_t as_create(ArtContext *c) {
    int32_t r = ensure_wide(c, true); if (r) return r;
    r = sync_all(c); if (r) return r;
    const auto t_begin = std::chrono::steady_clock::now();
    // (default: twice the frames in flight, 4 at least and 24 at most.  The host may issue the refit of frame n once the frames that read that version -- frame n - K -- are
    //  over, so K sets how far it runs ahead of the GPU: with K = F + 1 a refit is issued when its ring slot's previous frame has all but finished and its latency -- 0.5 ms
    //  among eight frames in flight -- stands in front of the slot's next frame; with K = 2 F it is a ring trip ahead.  Config 2, F = 8, a model of 164 k triangles moving
    //  every frame: 0.252 / 0.229 / 0.213 / 0.210 / 0.205 / 0.205 ms a frame with 4 / 8 / 10 / 12 / 16 / 24 versions (profiles/README.md round 4d); a version is the tree's
    //  arrays once more: 42 MB for config 2, 450 MB for config 4.)
    const uint32_t K = c->tuning.as_versions ? std::min(c->tuning.as_versions, kMaxAsVersions) : std::min(std::max(2u * c->F, 4u), kMaxAsVersions);
    const size_t np = c->h_dev_prims.size(), T = c->T, NW = c->bvh.n_wide;
    c->as.assign(K, AsVersion{});
    hipStream_t s = c->main_stream();
    double t_sec[6] = {0, 0, 0, 0, 0, 0};
    auto lap = [&, last = std::chrono::steady_clock::now()](int i) mutable { const auto n = std::chrono::steady_clock::now(); t_sec[i] += std::chrono::duration<double, std::milli>(n - last).count(); last = n; };
    auto body = [&]() -> int32_t {
        const uint32_t want = c->tuning.refit_streams == 0xFFFFFFFFu ? 0u : (c->tuning.refit_streams ? std::min(c->tuning.refit_streams, 4u) : std::min(c->F, 4u));
        while (c->n_refit_streams < want) {   // (kept for the life of the context)
            int lo = 0, hi = 0; HIPC(hipDeviceGetStreamPriorityRange(&lo, &hi));   // hi: the numerically smallest = the most urgent: a refit is a handful of small launches a whole frame waits for
            HIPC(hipStreamCreateWithPriority(&c->refit_stream[c->n_refit_streams], hipStreamNonBlocking, hi)); c->n_refit_streams++;
        }
        while (c->n_refit_streams > want) { c->n_refit_streams--; (void)hipStreamSynchronize(c->refit_stream[c->n_refit_streams]); (void)hipStreamDestroy(c->refit_stream[c->n_refit_streams]); c->refit_stream[c->n_refit_streams] = nullptr; }
        lap(0);
        if (!c->bvh.leaf_parent) { // who holds whom in the 4-wide tree: the marks of a refit go up along it
            HIPC(hipMalloc(&c->bvh.leaf_parent, T * 4)); HIPC(hipMalloc(&c->bvh.node_parent, NW * 4));
            launch_wide_parents(c->bvh.n_wide, c->bvh.widef, c->bvh.leaf_parent, c->bvh.node_parent, s);
            HIPC(hipGetLastError());
            hipError_t e = refit_lists_build(c->bvh, c->T, s);   // the refit's work lists (a workgroup per batch of subtrees)
            if (e != hipSuccess) return hipfail(e, "refit_lists_build");
        }
        lap(1);
        // one device block and one pinned block, carved per version (256-byte steps)
        auto pad = [](size_t n) { return (n + 255) & ~(size_t)255; };
        const size_t nbat = c->bvh.sub_batches ? c->bvh.sub_batches : 1;
        const size_t dev_owned = pad(T * sizeof(DevTri)) + pad(NW * sizeof(DevNodeW)) + pad(NW * sizeof(DevNode4)) + pad(np * sizeof(DevPrim)), dev_every = pad(NW * 4) + pad(32) + pad(nbat * 8);
        const size_t pin_every = pad(np * sizeof(DevPrim)) + pad(np) + pad(32) + pad(nbat * 4);
        HIPC(hipMalloc(&c->as_block, (K - 1) * dev_owned + K * dev_every));
        HIPC(hipHostMalloc(&c->as_pinned, K * pin_every, hipHostMallocDefault));
        lap(2);
        char *dp = (char *)c->as_block, *hp = (char *)c->as_pinned, *dhp = nullptr;
        HIPC(hipHostGetDevicePointer((void **)&dhp, c->as_pinned, 0));
        auto carve = [&](char *&p, size_t n) { char *q = p; p += pad(n); return q; };
        for (uint32_t v = 0; v < K; v++) { // (everything on the context's first stream, asynchronously: one wait at the end)
            AsVersion &V = c->as[v];
            if (v == 0) { V.tris = c->bvh.tris; V.widef = c->bvh.widef; V.wide = c->bvh.wide; V.prims = c->d_prims.p; }
            else {
                V.owned = true;
                V.tris = (DevTri *)carve(dp, T * sizeof(DevTri)); V.widef = (DevNodeW *)carve(dp, NW * sizeof(DevNodeW)); V.wide = (DevNode4 *)carve(dp, NW * sizeof(DevNode4)); V.prims = (DevPrim *)carve(dp, np * sizeof(DevPrim));
                HIPC(hipMemcpyAsync(V.tris, c->bvh.tris, T * sizeof(DevTri), hipMemcpyDeviceToDevice, s)); HIPC(hipMemcpyAsync(V.widef, c->bvh.widef, NW * sizeof(DevNodeW), hipMemcpyDeviceToDevice, s));
                HIPC(hipMemcpyAsync(V.wide, c->bvh.wide, NW * sizeof(DevNode4), hipMemcpyDeviceToDevice, s)); HIPC(hipMemcpyAsync(V.prims, c->d_prims.p, np * sizeof(DevPrim), hipMemcpyDeviceToDevice, s));
            }
            V.mark = (uint32_t *)carve(dp, NW * 4); V.acc = (double *)carve(dp, 32); V.batch_cost = (double *)carve(dp, nbat * 8);
            HIPC(hipMemsetAsync(V.mark, 0, NW * 4, s)); HIPC(hipMemsetAsync(V.acc, 0, 32, s));
            const size_t off = (size_t)(hp - (char *)c->as_pinned);
            V.h_prims = (DevPrim *)carve(hp, np * sizeof(DevPrim)); V.h_touched = (uint8_t *)carve(hp, np); V.h_result = (double *)carve(hp, 32);
            V.dh_prims = (DevPrim *)(dhp + off); V.dh_touched = (uint8_t *)(dhp + off + pad(np * sizeof(DevPrim))); V.dh_result = (double *)(dhp + off + pad(np * sizeof(DevPrim)) + pad(np));
            V.h_dirty = (uint32_t *)carve(hp, nbat * 4); V.dh_dirty = (uint32_t *)(dhp + off + pad(np * sizeof(DevPrim)) + pad(np) + pad(32));
            HIPC(hipEventCreateWithFlags(&V.ready, hipEventDisableTiming));
        }
        lap(3);
        AsVersion &V0 = c->as[0];
        launch_wide_cost(c->bvh.n_wide, V0.widef, nullptr, V0.acc, V0.dh_result, s);   // the cost of the tree as built
        HIPC(hipGetLastError()); HIPC(hipStreamSynchronize(s));
        c->as_cost0 = V0.h_result[0]; c->refit_cost_ratio = 1.0f;
        lap(4);
        if (c->tuning.log & 1u) std::fprintf(stderr, "[art] versions: %u of them; refit streams %.2f ms, parents + work lists %.2f, the two allocations %.2f, copies issued + events %.2f, the wait for them + the cost of the tree as built %.2f\n", K, t_sec[0], t_sec[1], t_sec[2], t_sec[3], t_sec[4]);
        return ART_OK;
    };
    r = body();
    if (r) as_release(c);
    c->versions_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return r;
}
// the surface-area cost of the latest refit travels to the host behind it; once it has arrived it is what ArtStats.refit_cost_ratio and the rebuild rule go by
void harvest_cost(ArtContext *c) {
    if (c->as.empty()) return;
    AsVersion &L = c->as[c->as_cur];
    if (!L.result_pending || hipEventQuery(L.ready) != hipSuccess) return;
    L.result_pending = false; L.ready_known = true;
    if (c->as_cost0 > 0.0) c->refit_cost_ratio = (float)(L.h_result[0] / c->as_cost0);
    const unsigned long long *st = reinterpret_cast<const unsigned long long *>(L.h_result);
    c->last_refit_ms = (float)((double)(st[3] - st[2]) * 1e-5);   // wall_clock64: 100 MHz
}
// A model moved since the last launch: bring the NEXT version of the structure up to date on stream s, the stream of ring slot k whose frame is about to be
// launched -- the frame is ordered behind the refit by the stream, frames on other streams by V.ready (art_trace).  Frames still reading the version about to be
// written are waited for on the host, like the reference's per-frame fence (renderer.rs:451-466).
// What the stream sees: three launches and one event record (round 3: two uploads, a launch per tree level, a clear, the cost's read-back and four event records --
// twenty operations, 0.3 ms of issue in front of a 0.1 ms refit).
int32_t scene_refresh(ArtContext *c, uint32_t k, hipStream_t s) {
    if (!c->xform_dirty) return ART_OK;
    int32_t r;
    if (c->as.empty()) { // (a host that announced its moves with ART_FLAG_DYNAMIC_SCENE paid this in art_scene_build)
        const auto t_begin = std::chrono::steady_clock::now();
        r = as_create(c); if (r) return r;
        c->first_move_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    }
    {   // the cost of the latest refit, if it has arrived: past the threshold the tree is built again for where the models are now
        harvest_cost(c);
        const float thr = c->tuning.refit_rebuild_ratio > 0.0f ? c->tuning.refit_rebuild_ratio : (c->tuning.refit_rebuild_ratio < 0.0f ? INFINITY : 2.0f);
        if (c->refit_cost_ratio > thr) {
            if (c->tuning.log & 1u) std::fprintf(stderr, "[art] refit cost %.2f x the build's: building again\n", c->refit_cost_ratio);
            r = art_scene_build(c); // (synchronises, uploads the primitives with their current matrices, drops the versions)
            if (r == ART_OK) c->rebuilds++;
            return r;
        }
    }
    const uint32_t K = (uint32_t)c->as.size(), next = (c->as_cur + 1) % K;
    AsVersion &V = c->as[next];
    const bool beside = c->n_refit_streams != 0;   // the refit runs beside the slot's frames, on a stream of its own
    if (beside) s = c->refit_stream[k % c->n_refit_streams];
    for (uint32_t j = 0; j < c->F; j++) {
        if (j != k || beside) { // (on the frame's own stream ring slot k's earlier work is ordered before the refit by that stream)
            if (V.aux[j]) HIPC(hipStreamSynchronize(c->stream_of(j)));
            else if (V.used[j]) {
                const uint64_t f = V.used[j] - 1;
                if (c->frame_no - f <= (uint64_t)ArtContext::kRing) HIPC(hipEventSynchronize(c->ev[f % ArtContext::kRing][4])); else HIPC(hipStreamSynchronize(c->stream_of(j)));
            }
        }
        V.used[j] = 0; V.aux[j] = false;
    }
    // the staging memory below is read by the kernels of the refit that wrote this version last: that refit has to be over before the host writes it again (it is, whenever the
    // frames above were waited for -- they ran behind it -- but nothing else says so: a version no frame ever read, a ring slot that skipped its turn)
    if (!V.ready_known) { HIPC(hipEventSynchronize(V.ready)); V.ready_known = true; }
    V.result_pending = false;
    if (c->graph_mode) drop_graphs(c); // a captured frame holds the old version's pointers
    const size_t np = c->h_dev_prims.size();
    std::memcpy(V.h_prims, c->h_dev_prims.data(), np * sizeof(DevPrim));
    for (size_t p = 0; p < np; p++) V.h_touched[p] = (p < c->prim_moved.size() && c->prim_moved[p] > V.epoch) ? 1 : 0;   // what moved since THIS version was written (it may be several refits behind)
    RefitArgs ra{};
    ra.T = c->T; ra.n_wide = c->bvh.n_wide; ra.n_prims = (uint32_t)np; ra.shade = c->bvh.shade_tris; ra.prims_host = V.dh_prims; ra.prims_dev = V.prims; ra.touched = V.dh_touched;
    ra.sub_nodes = c->bvh.sub_nodes; ra.sub_leaves = c->bvh.sub_leaves; ra.sub_off = c->bvh.sub_off; ra.sub_batches = c->bvh.sub_batches; ra.sub_levels = c->bvh.sub_levels;
    ra.leaf_parent = c->bvh.leaf_parent; ra.node_parent = c->bvh.node_parent; ra.mark = V.mark; ra.tris = V.tris; ra.wide = V.wide; ra.widef = V.widef; ra.acc = V.acc; ra.result = V.dh_result;
    {   // the batches that hold a primitive that moved (since this version was written); the others keep their triangles, their boxes and -- in a large tree -- their
        // cached share of the cost, which a version's first refit makes for all of them
        const std::vector<uint32_t> &po = c->bvh.batch_prim_off, &pi = c->bvh.batch_prim_ids;
        ra.fold = c->bvh.n_wide >= (c->tuning.refit_fold_nodes ? c->tuning.refit_fold_nodes : kFoldRequantNodes);
        const bool all = po.size() != (size_t)c->bvh.sub_batches + 1 || (ra.fold && !V.cost_cached);
        uint32_t nd = 0;
        if (!all) for (uint32_t b = 0; b < c->bvh.sub_batches; b++) {
            bool hit = false;
            for (uint32_t i = po[b]; i < po[b + 1] && !hit; i++) hit = pi[i] < np && V.h_touched[pi[i]] != 0;
            if (hit) V.h_dirty[nd++] = b;
        }
        ra.dirty = all ? nullptr : V.dh_dirty; ra.n_dirty = nd; ra.batch_cost = V.batch_cost;
        if (all) V.cost_cached = true;
    }
    launch_refit(ra, s);
    HIPC(hipEventRecord(V.ready, s)); V.ready_known = false; V.ready_slot = beside ? ~0u : k; V.result_pending = true;   // (~0: no frame stream is behind it by itself)
    HIPC(hipGetLastError());
    c->as_cur = next; V.epoch = ++c->as_epoch; c->xform_dirty = false; c->refits++;
    return ART_OK;
}
// for the calls that read the structure outside a frame (queries, the parity surface): nothing in flight, the pending move applied
int32_t refresh_now(ArtContext *c) {
    int32_t r = sync_all(c); if (r) return r;
    if (!c->xform_dirty) return ART_OK;
    r = scene_refresh(c, 0, c->stream_of(0)); if (r) return r;
    r = sync_all(c); if (r) return r;   // (the refit may have run on a stream of its own)
    if (!c->as.empty()) c->as[c->as_cur].ready_known = true;
    return ART_OK;
}
// the binary trees and the 64-byte node records follow the versions on demand only (the non-default walks and the parity surface read them): they are
// not versioned, so this waits for everything in flight
int32_t ensure_binary(ArtContext *c, bool needed) {
    if (!needed || c->binary_epoch == as_epoch_of(c, c->as_cur)) return ART_OK;
    int32_t r = sync_all(c); if (r) return r;
    hipError_t e = binary_refit(c->bvh, c->T, as_ptrs(c, c->as_cur).tris, c->main_stream());
    if (e != hipSuccess) return hipfail(e, "binary_refit");
    c->binary_epoch = as_epoch_of(c, c->as_cur);
    drop_graphs(c);
    return ART_OK;
}


// ---- wave plan of the fused frame ----------------------------------------------------------------------------------------------------
// the first table of a frame layout: every 8x8 block one wave, in the XCD-aware launch order (k_plan writes the later ones in the same layout)
static void plan_first_items(const WavePlan &P, uint32_t n64, std::vector<uint2> &out) {
    out.clear();
    for (uint32_t blk : P.order)
        for (uint32_t w = 0; w < 4; w++) out.push_back(make_uint2(blk * 4 + w, 0xFFFFu));
    (void)n64;
    // k_frame runs one wave per workgroup, and workgroup j lands on XCD j % 8 (round-robin dispatch): deal the items so that the four waves of launch
    // block 8g + x (a 256-pixel block the XCD-aware order gave to XCD x) stay on XCD x -- positions 32g + 8k + x, k = 0..3.  A permutation whatever the
    // hardware does; only the L2 locality depends on it.
    std::vector<uint2> q(out);
    for (size_t g = 0; (g + 1) * 32 <= out.size(); g++)
        for (uint32_t x = 0; x < 8; x++) for (uint32_t k = 0; k < 4; k++) q[g * 32 + 8 * k + x] = out[g * 32 + 4 * x + k];
    out.swap(q);
}
static int32_t plan_reset(ArtContext *c) {
    WavePlan &P = c->plan;
    const ArtTuning &t = c->tuning;
    P.enabled = !t.fixed_waves && !(c->cfg.flags & ART_FLAG_FIXED_WAVES);
    P.min_steps = t.split_min_steps ? t.split_min_steps : 150;
    P.fixed_steps = t.split_fixed_steps;
    P.alpha = t.split_alpha > 0.f ? t.split_alpha : 0.7f;
    const uint32_t hwq = t.hw_queues ? t.hw_queues : 4;   // HIP's default number of hardware queues per process; a host that raises GPU_MAX_HW_QUEUES says so in ArtTuning
    // (at most 4: a ring of 8 hides a straggler while it stays full, but a run's last frames drain without neighbours -- over the driver's 20 steps a plan made for 3-4 launches
    //  in flight is worth 4 %, over 1 000 steps it costs 1 %: profiles/README.md round 4)
    P.in_flight = std::max(1u, std::min(std::min(c->F, hwq), 4u));
    const uint32_t n64 = c->n_local / 64;
    P.cap = n64 + n64 / 2 + 64;       // at most half as many waves again
    std::vector<uint2> first;
    plan_first_items(P, n64, first);
    for (int i = 0; i < 2; i++) { HIPC(P.d_items[i].ensure(P.cap)); P.n_items[i] = 0; P.retire_set[i] = false; }
    if (!first.empty()) HIPC(hipMemcpy(P.d_items[0].p, first.data(), first.size() * sizeof(uint2), hipMemcpyHostToDevice));
    P.n_items[0] = (uint32_t)first.size(); P.cur = 0; P.split1 = P.split2 = 0;
    HIPC(P.d_level.ensure(n64 ? n64 : 1)); HIPC(P.d_level_tmp.ensure(n64 ? n64 : 1)); HIPC(P.d_worst.ensure(n64 ? n64 : 1));
    HIPC(hipMemset(P.d_level.p, 0, n64 ? n64 : 1)); HIPC(hipMemset(P.d_worst.p, 0, (size_t)(n64 ? n64 : 1) * 4));
    if (!P.h_result) { HIPC(hipHostMalloc((void **)&P.h_result, 32, hipHostMallocDefault)); HIPC(hipHostGetDevicePointer((void **)&P.dh_result, P.h_result, 0)); }
    std::memset(P.h_result, 0, 32);
    if (!P.cost_ready) HIPC(hipEventCreateWithFlags(&P.cost_ready, hipEventDisableTiming));
    if (!P.plan_stream) { int lo = 0, hi = 0; HIPC(hipDeviceGetStreamPriorityRange(&lo, &hi)); HIPC(hipStreamCreateWithPriority(&P.plan_stream, hipStreamNonBlocking, lo)); }   // (the least urgent: nothing waits for it)
    else HIPC(hipStreamSynchronize(P.plan_stream));
    P.pending = false; P.next_sample = c->frame_no; P.interval = 1; P.replans = 0; P.last_sample = c->frame_no;
    return ART_OK;
}
// the table the current one alternates with: free once every launch that read it has finished (the events recorded on all frame streams when it was left)
static bool plan_other_free(ArtContext *c) {
    WavePlan &P = c->plan;
    const int other = P.cur ^ 1;
    if (!P.retire_set[other]) return true;
    for (uint32_t k = 0; k < c->F; k++) if (hipEventQuery(P.retire[other][k]) != hipSuccess) return false;
    return true;
}
// behind a sampled frame, on its stream: the next plan from what its waves counted (k_plan writes the other table if a level changed that matters)
static int32_t plan_launch(ArtContext *c, const FrameArgs &a, hipEvent_t frame_done) {
    WavePlan &P = c->plan;
    hipStream_t s = P.plan_stream;
    HIPC(hipStreamWaitEvent(s, frame_done, 0));   // (a wait in the PLAN's stream: the frames' streams see nothing of it)
    PlanArgs pa{};
    pa.items_in = P.d_items[P.cur].p; pa.n_items_in = a.n_wave_items; pa.cost = a.wave_cost;
    pa.level = P.d_level.p; pa.level_tmp = P.d_level_tmp.p; pa.n64 = c->n_local / 64; pa.worst = P.d_worst.p;
    pa.order = c->d_block_order.p; pa.n256 = c->n_local / 256;
    pa.items_out = P.d_items[P.cur ^ 1].p; pa.cap = P.cap;
    constexpr float kWaveSlots = 256.0f * 32.0f; // CUs x waves per CU
    pa.share = P.alpha * (float)c->B * (float)P.in_flight / kWaveSlots;   // (the counts are one frame's; a launch traces B frames)
    pa.min_steps = P.min_steps; pa.fixed_steps = P.fixed_steps; pa.result = P.dh_result;
    launch_plan(pa, s);
    HIPC(hipEventRecord(P.cost_ready, s));
    P.pending = true; P.pending_table = P.cur; P.last_sample = c->frame_no;
    return ART_OK;
}
// The view or the lights changed: the heavy blocks are elsewhere, sooner or later.  The plan in use stays (a camera that moves like the reference's -- 0.002 units per
// millisecond, main.rs:80-105 -- shifts them by a fraction of a pixel a frame) and the waves are looked at again within kMovingInterval frames of the last look: a camera
// that moves every frame is sampled at that cadence, not at every frame (round 3 reset the interval to 1 here: every frame that found no sample in flight was a counting
// frame and every other poll a new table).
constexpr uint32_t kMovingInterval = 32;
static uint32_t plan_moving_interval(const ArtContext *c) { return c->tuning.plan_moving_interval ? c->tuning.plan_moving_interval : kMovingInterval; }
static void plan_hint_moved(ArtContext *c) {
    WavePlan &P = c->plan;
    const uint32_t mi = plan_moving_interval(c);
    P.interval = std::min(P.interval, mi);
    P.next_sample = std::min<uint64_t>(P.next_sample, P.last_sample + mi);
    P.moved_since_poll = true;
}

// A sampled frame's plan has been made: if it wrote a new table, that one becomes the current one (the table being left stays in use until every frame stream has passed this point).
static int32_t plan_poll(ArtContext *c) {
    WavePlan &P = c->plan;
    if (!P.pending || hipEventQuery(P.cost_ready) != hipSuccess) return ART_OK;
    P.pending = false;
    const uint32_t *res = P.h_result;
    const int verbose = (c->tuning.log & 4u) ? 2 : ((c->tuning.log & 2u) ? 1 : 0);
    if (verbose > 1) std::fprintf(stderr, "[art] plan poll at frame %llu: table %d sampled, %s, slowest wave %u steps, target %u, interval %u\n", (unsigned long long)c->frame_no, P.pending_table, res[1] ? "a new table" : "the table stays", res[5], res[4], P.interval);
    if (res[1]) {
        for (uint32_t k = 0; k < c->F; k++) {
            if (!P.retire[P.cur][k]) HIPC(hipEventCreateWithFlags(&P.retire[P.cur][k], hipEventDisableTiming));
            HIPC(hipEventRecord(P.retire[P.cur][k], c->stream_of(k)));
        }
        P.retire_set[P.cur] = true;
        P.cur ^= 1; P.n_items[P.cur] = res[0]; P.split1 = res[2]; P.split2 = res[3]; P.replans++;
        if (verbose) std::fprintf(stderr, "[art] wave plan %u at frame %llu: %u blocks in 4, %u in 16, of %u; slowest sampled wave %u steps, target %u\n", P.replans, (unsigned long long)c->frame_no, res[2], res[3], c->n_local / 64, res[5], res[4]);
        P.interval = P.moved_since_poll ? plan_moving_interval(c) : c->F + 1;        // a still view: let frames of the new plan come back, then judge it; a moving one: at its cadence
    } else P.interval = P.interval < 128 ? P.interval * 2 : 256;
    P.moved_since_poll = false;
    P.next_sample = c->frame_no + P.interval;
    return ART_OK;
}

// A directional light's L vector, its length and the shadow ray's reciprocal direction are the same for every pixel (light.glsl:96: -dir * 10): the
// kernel-argument copy of such a record carries them in fields a directional light does not use (area_pos2 = L, penumbra_angle = |nn_L|, area_pos3 =
// 1 / safe(L)), made here with the operations the kernel would run per lane -- correctly rounded sqrt and division, explicit fma, nothing contracted
// (the library is built with -ffp-contract=off): the same bits.  The caller's records are untouched.
static void directional_constants(ArtLight &l) {
    if (l.type != 2u) return;
    const float nx = -l.dir[0] * 10.0f, ny = -l.dir[1] * 10.0f, nz = -l.dir[2] * 10.0f;           // neg(dir) * 10
    const float d = std::fmaf(nz, nz, std::fmaf(ny, ny, nx * nx));                               // dot3
    const float len = std::sqrt(d), inv = 1.0f / std::sqrt(d);
    const float L[3] = {nx * inv, ny * inv, nz * inv};                                           // nrm3
    for (int k = 0; k < 3; k++) {
        l.area_pos2[k] = L[k];
        const float sd = std::fabs(L[k]) < 1e-20f ? std::copysign(1e-20f, L[k]) : L[k];          // safe_dir
        l.area_pos3[k] = 1.0f / sd;
    }
    l.penumbra_angle = len;
}

int32_t setup_frame(ArtContext *c) {
    // tile ownership + per-frame buffers for the current extent / light count
    c->tiles_x = (c->W + kTile - 1) / kTile; c->tiles_y = (c->H + kTile - 1) / kTile;
    uint32_t count = c->cfg.shard_count > 1 ? c->cfg.shard_count : 1, rank = count > 1 ? c->cfg.shard_rank : 0;
    c->tile_list.clear();
    std::vector<uint32_t> per(count, 0), slot_of((size_t)c->tiles_x * c->tiles_y);
    const std::vector<uint8_t> owner_of = shard_owner_table(c->tiles_x, c->tiles_y, count, c->cfg.root_relief);
    for (uint32_t ty = 0; ty < c->tiles_y; ty++)
        for (uint32_t tx = 0; tx < c->tiles_x; tx++) {
            uint32_t o = owner_of[(size_t)ty * c->tiles_x + tx];
            slot_of[(size_t)ty * c->tiles_x + tx] = (o << 24) | per[o];
            per[o]++;
            if (o == rank) c->tile_list.push_back(ty * c->tiles_x + tx);
        }
    HIPC(c->d_tile_slot.ensure(slot_of.size()));
    HIPC(hipMemcpy(c->d_tile_slot.p, slot_of.data(), slot_of.size() * 4, hipMemcpyHostToDevice));
    c->padded_tiles = 0;
    for (uint32_t v : per) c->padded_tiles = v > c->padded_tiles ? v : c->padded_tiles;
    c->n_local = (uint32_t)c->tile_list.size() * kTilePixels;
    size_t npix = (size_t)c->W * c->H;
    size_t nl = c->lights.size() ? c->lights.size() : 1;
    HIPC(c->d_tile_list.ensure(c->tile_list.size()));
    if (!c->tile_list.empty()) HIPC(hipMemcpy(c->d_tile_list.p, c->tile_list.data(), c->tile_list.size() * 4, hipMemcpyHostToDevice));
    {
        std::vector<uint32_t> xy(c->tile_list.size());
        for (size_t i = 0; i < xy.size(); i++) xy[i] = (c->tile_list[i] % c->tiles_x) | ((c->tile_list[i] / c->tiles_x) << 16);
        HIPC(c->d_tile_xy.ensure(xy.size()));
        if (!xy.empty()) HIPC(hipMemcpy(c->d_tile_xy.p, xy.data(), xy.size() * 4, hipMemcpyHostToDevice));
    }
    {   // Workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MB L2.  Group the owned tiles into macro-blocks of
        // kMacro x kMacro tiles, deal the macro-blocks round-robin to the XCDs (balance: every XCD gets pieces from all over the
        // frame) and order the launch so that XCD x works through ITS macro-blocks: its L2 then holds the BVH of a few screen
        // regions instead of the whole view, in every kernel of the frame and in every frame in flight.  A permutation of the
        // blocks whatever the hardware's dispatch order is; only the locality depends on it.  (ArtTuning.block_order 1: identity.)
        const uint32_t macro = c->macro;
        const uint32_t nb = c->n_local / 256;
        std::vector<uint32_t> order(nb);
        for (uint32_t b = 0; b < nb; b++) order[b] = b;
        if (macro > 0 && nb >= 64) {
            const uint32_t mx = (c->tiles_x + macro - 1) / macro;
            std::vector<std::vector<uint32_t>> queue(8);
            std::vector<std::pair<uint32_t, uint32_t>> keyed; // (macro-block id, block)
            for (uint32_t b = 0; b < nb; b++) { uint32_t t = c->tile_list[b >> 2]; keyed.push_back({(t / c->tiles_x / macro) * mx + (t % c->tiles_x) / macro, b}); }
            std::stable_sort(keyed.begin(), keyed.end());
            uint32_t rank = 0;
            for (size_t i = 0; i < keyed.size(); i++) { if (i && keyed[i].first != keyed[i - 1].first) rank++; queue[rank & 7u].push_back(keyed[i].second); }
            // launch position b runs on XCD b % 8: take that XCD's next block; once a queue is empty its positions take what is left
            size_t head[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t b = 0; b < nb; b++) {
                uint32_t q = b & 7u;
                for (uint32_t tries = 0; tries < 8 && head[q] >= queue[q].size(); tries++) q = (q + 1) & 7u;
                order[b] = queue[q][head[q]++];
            }
        }
        HIPC(c->d_block_order.ensure(nb ? nb : 1));
        if (nb) HIPC(hipMemcpy(c->d_block_order.p, order.data(), (size_t)nb * 4, hipMemcpyHostToDevice));
        c->plan.order = order;
    }
    {
        int32_t pr = plan_reset(c); if (pr) return pr;
    }
    for (uint32_t k = 0; k < c->F; k++) {
        FrameSlot &S = c->slot[k];
        HIPC(S.d_wave_cost.ensure(c->plan.cap ? c->plan.cap : 1));
        HIPC(S.d_counters.ensure(kCounterWords)); HIPC(hipMemset(S.d_counters.p, 0, kCounterWords * 4)); // packet frames keep them clear themselves (k_accumulate)
        const bool staged = !(c->fused && c->kind_primary == 8 && c->kind_shadow == 8); // the fused frame keeps these records in registers
        const size_t B = c->B; // frames per launch: every output holds B frames back to back
        if (staged || (c->cfg.flags & ART_FLAG_KEEP_DEBUG)) HIPC(S.d_hits.ensure(c->n_local * B));
        if (staged) { HIPC(S.d_contrib.ensure(nl * c->n_local)); HIPC(S.d_shadow_rays.ensure(2 * nl * c->n_local)); }
        HIPC(S.d_color.ensure(npix * B)); HIPC(S.d_normal.ensure(npix * B)); HIPC(S.d_depth.ensure(npix * B));
        HIPC(hipMemset(S.d_color.p, 0, npix * B * 16)); HIPC(hipMemset(S.d_normal.p, 0, npix * B * 16)); HIPC(hipMemset(S.d_depth.p, 0, npix * B * 4));
        if (c->tiled()) { HIPC(S.d_color_tiles.ensure((size_t)c->padded_tiles * kTilePixels * B)); HIPC(hipMemset(S.d_color_tiles.p, 0, (size_t)c->padded_tiles * kTilePixels * B * c->tile_px_bytes())); }
        if ((c->cfg.flags & ART_FLAG_KEEP_DEBUG) || c->fused) HIPC(S.d_shadow_bits.ensure(c->n_local * B)); // fused frames always write their per-pixel shadow bits (stats)
        if (c->fused && c->lights.size() > (size_t)kMaxLights) HIPC(S.d_pix_more.ensure(c->n_local * B));
        if (S.ext_tiles && S.ext_tiles_bytes != (size_t)c->padded_tiles * kTilePixels * c->tile_px_bytes() * B) { S.ext_tiles = nullptr; S.ext_ring_n = 0; S.tiles_of_last = nullptr; S.ext_tiles_bytes = 0; }
    }
    HIPC(hipDeviceSynchronize()); // the clears above ran on the null stream; the slots' streams are non-blocking
    drop_graphs(c);
    c->frame_ready = true;
    return ART_OK;
}

void normalize3(const float *v, float *o) {
    float l = std::sqrt(std::fmaf(v[2], v[2], std::fmaf(v[1], v[1], v[0] * v[0])));
    float inv = 1.0f / l;
    o[0] = v[0] * inv; o[1] = v[1] * inv; o[2] = v[2] * inv;
}
void cross3h(const float *a, const float *b, float *o) {
    o[0] = std::fmaf(a[1], b[2], -(a[2] * b[1])); o[1] = std::fmaf(a[2], b[0], -(a[0] * b[2])); o[2] = std::fmaf(a[0], b[1], -(a[1] * b[0]));
}
float dot3h(const float *a, const float *b) { return std::fmaf(a[2], b[2], std::fmaf(a[1], b[1], a[0] * b[0])); }

// general 4x4 inverse by cofactors, column-major (stands in for nalgebra's try_inverse, vk_camera.rs:111-113)
bool mat4_inverse(const float *m, float *o) {
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
    if (!(std::fabs(det) > 0.0f)) return false; // singular, or NaN somewhere in m
    det = 1.0f / det;
    bool finite = true;
    for (int i = 0; i < 16; i++) { o[i] = inv[i] * det; finite = finite && std::isfinite(o[i]); }
    return finite;
}

} // namespace

// a camera block with a NaN or an infinity in it makes rays no triangle can be tested against (the packed struct is read through a copy: its floats are unaligned)
static bool camera_finite(const ArtCamera *cam) {
    float v[sizeof(ArtCamera) / 4];
    std::memcpy(v, cam, sizeof(ArtCamera));
    for (float x : v) if (!std::isfinite(x)) return false;
    return true;
}

// hit records name a triangle by its global id: the primitive is the last slot whose first triangle is <= gid (k_soup's rule)
static void gid_to_ids(const ArtContext *c, uint32_t gid, int32_t *ids) {
    const std::vector<uint32_t> &f = c->h_first_tri;
    size_t lo = 0, hi = f.size();
    while (hi - lo > 1) { size_t mid = (lo + hi) >> 1; if (f[mid] <= gid) lo = mid; else hi = mid; }
    ids[0] = (int32_t)lo; ids[1] = (int32_t)(gid - f[lo]);
}

int32_t art::ring_rewind(ArtContext *c) {
    if (!c) return fail(ART_E_INVALID, "ring_rewind: null context");
    int32_t r = use_device(c); if (r) return r;
    r = sync_all(c); if (r) return r;
    c->frame_no = 0; c->collected_upto = 0; c->last = 0;
    c->plan.next_sample = 0; c->plan.pending = false; c->plan.last_sample = 0;   // (a sample in flight has landed: everything is synchronised)
    return ART_OK;
}

extern "C" {

const char *art_last_error(void) { return g_err.c_str(); }

int32_t art_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t art_create(const ArtConfig *cfg, ArtContext **out) {
    if (!cfg || !out) return fail(ART_E_INVALID, "art_create: null argument");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(ART_E_NO_DEVICE, "art_create: no HIP device (libart has no CPU fallback)");
    int dev = cfg->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) return fail(ART_E_NO_DEVICE, "art_create: hipGetDevice failed"); }
    if (dev >= n) return fail(ART_E_INVALID, "art_create: device ordinal out of range");
    hipDeviceProp_t prop;
    HIPC(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ART_E_NO_DEVICE, std::string("art_create: device is ") + prop.gcnArchName + ", libart is built for gfx950 only");
    if (cfg->morton_bits != 0 && cfg->morton_bits != 30 && cfg->morton_bits != 63) return fail(ART_E_INVALID, "art_create: morton_bits must be 0, 30 or 63");
    if (cfg->shard_count > 1 && cfg->shard_rank >= cfg->shard_count) return fail(ART_E_INVALID, "art_create: shard_rank >= shard_count");
    if (cfg->shard_count > 255) return fail(ART_E_INVALID, "art_create: at most 255 shards");
    if (cfg->frames_in_flight > kMaxFrames) return fail(ART_E_INVALID, "art_create: at most 24 frames in flight");
    if (cfg->root_relief > 255) return fail(ART_E_INVALID, "art_create: root_relief 0..255");
    ArtContext *c = new (std::nothrow) ArtContext();
    if (!c) return fail(ART_E_NOMEM, "art_create: out of memory");
    c->cfg = *cfg;
    if (c->cfg.morton_bits == 0) c->cfg.morton_bits = 63;
    c->device = dev;
    c->F = cfg->frames_in_flight == 0 ? 1 : cfg->frames_in_flight;
    hipError_t e = hipSetDevice(dev);
    for (uint32_t k = 0; k < c->F && e == hipSuccess; k++) {
        e = acquire_stream(c->device, &c->slot[k].own);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->slot[k].done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreate(&c->slot[k].ao_ev[0]);
        if (e == hipSuccess) e = hipEventCreate(&c->slot[k].ao_ev[1]);
    }
    for (int f = 0; f < ArtContext::kRing && e == hipSuccess; f++)
        for (int i = 0; i < 5 && e == hipSuccess; i++) e = hipEventCreate(&c->ev[f][i]);
    if (e != hipSuccess) { (void)art_destroy(c); return hipfail(e, "art_create"); } // releases the streams and events created so far
    c->W = cfg->width; c->H = cfg->height;
    // the fused packet frame is the default at every ring depth (one frame at a time: 0.575 ms against 0.669 ms for the staged per-ray
    // kernels, profiles/README.md r1h); art_set_tuning selects the other forms
    c->fast_trace = !(cfg->flags & ART_FLAG_FAST_BUILD);
    build_prewarm(c->main_stream()); sah_prewarm(c->main_stream());   // the builders' code objects are loaded here, once a process, not inside the first art_scene_build
    (void)hipGetLastError();
    *out = c;
    return ART_OK;
}

int32_t art_destroy(ArtContext *c) {
    if (!c) return ART_OK;
    (void)hipSetDevice(c->device);
    for (uint32_t k = 0; k < c->F; k++) if (c->stream_of(k)) (void)hipStreamSynchronize(c->stream_of(k));
    drop_graphs(c);
    for (uint32_t i = 0; i < c->n_refit_streams; i++) if (c->refit_stream[i]) { (void)hipStreamSynchronize(c->refit_stream[i]); (void)hipStreamDestroy(c->refit_stream[i]); }
    as_release(c);
    lbvh_free(c->bvh); c->arena.release();
    c->d_verts.release(); c->d_indices.release(); c->d_tex.release(); c->d_prims.release(); c->d_first_tri.release(); c->d_ao_tab.release();
    c->d_tile_list.release(); c->d_tile_xy.release(); c->d_tile_slot.release(); c->d_block_order.release(); c->plan.release();
    for (uint32_t k = 0; k < kMaxFrames; k++) {
        c->slot[k].release();
        if (c->slot[k].done) (void)hipEventDestroy(c->slot[k].done);
        for (int i = 0; i < 2; i++) if (c->slot[k].ao_ev[i]) (void)hipEventDestroy(c->slot[k].ao_ev[i]);
        if (c->slot[k].own) release_stream(c->device, c->slot[k].own);
    }
    for (int f = 0; f < ArtContext::kRing; f++)
        for (int i = 0; i < 5; i++) if (c->ev[f][i]) (void)hipEventDestroy(c->ev[f][i]);
    for (int i = 0; i < 2; i++) if (c->mark[i]) (void)hipEventDestroy(c->mark[i]);
    delete c;
    return ART_OK;
}

int32_t art_set_tuning(ArtContext *c, const ArtTuning *t) {
    if (!c || !t) return fail(ART_E_INVALID, "art_set_tuning: null argument");
    auto walk_ok = [](uint32_t k) { return k == 0 || k == 2 || k == 4; };
    if ((t->frame_form != 0 && t->frame_form != 2) || t->tree_builder > 1 || t->packet_wide > 2 || !walk_ok(t->primary_walk) || !walk_ok(t->shadow_walk) || !(walk_ok(t->ao_walk) || t->ao_walk == 6))
        return fail(ART_E_INVALID, "art_set_tuning: frame_form 0|2, tree_builder 0..1, packet_wide 0..2, walks 0|2|4");
    if (t->frame_form == 0 && (t->primary_walk || t->shadow_walk)) return fail(ART_E_INVALID, "art_set_tuning: the fused frame's rays are packets (primary_walk / shadow_walk choose the per-ray walks of frame_form 2)");
    if (t->as_versions > kMaxAsVersions || !(t->refit_rebuild_ratio == t->refit_rebuild_ratio)) return fail(ART_E_INVALID, "art_set_tuning: as_versions 0..24, refit_rebuild_ratio a number");
    int32_t r = use_device(c); if (r) return r;
    r = sync_all(c); if (r) return r;
    drop_graphs(c);
    c->tuning = *t;
    c->fused = t->frame_form == 0;
    c->kind_primary = t->frame_form == 2 ? 2 : 8; c->kind_shadow = t->frame_form == 2 ? 4 : 8; c->kind_ao = 4;   // per-ray frames: binary nodes for primary rays, 4-wide for shadow rays
    if (t->primary_walk) c->kind_primary = (int)t->primary_walk;
    if (t->shadow_walk) c->kind_shadow = (int)t->shadow_walk;
    if (t->ao_walk) c->kind_ao = (int)t->ao_walk;
    c->fast_trace = !(c->cfg.flags & ART_FLAG_FAST_BUILD);
    c->tree_builder = t->tree_builder == 1 ? 1 : 3;
    c->packet_wide = t->packet_wide != 2;   // 0: the default (4-wide), 1: 4-wide, 2: binary
    c->macro = t->block_order == 0 ? 2u : (t->block_order == 1 ? 0u : t->block_order);
    c->ao_entry = t->ao_entry_off == 0;
    c->wide_on_host = t->wide_builder == 1;
    c->built = false; c->frame_ready = false; c->traced = false;   // the tree and the frame layout are made again with the new choices
    return ART_OK;
}

int32_t art_set_stream(ArtContext *c, void *hip_stream) {
    if (!c) return fail(ART_E_INVALID, "art_set_stream: null context");
    if (hip_stream && c->F > 1) return fail(ART_E_STATE, "art_set_stream: a context with several frames in flight owns its streams (use art_stream_wait_frame / art_wait_external_event)");
    int32_t r = use_device(c); if (r) return r;
    r = sync_all(c); if (r) return r;
    c->ext_stream = (hipStream_t)hip_stream;
    return ART_OK;
}

int32_t art_scene_add_primitive(ArtContext *c, const ArtVertex *verts, uint32_t n_verts, const void *indices, uint32_t n_indices,
                                uint32_t idx_bytes, const uint8_t *rgba8, uint32_t tw, uint32_t th, const float model3x4[12], uint32_t *out_id) {
    if (!c || !verts || !indices || !rgba8 || !model3x4) return fail(ART_E_INVALID, "art_scene_add_primitive: null argument");
    if (idx_bytes != 2 && idx_bytes != 4) return fail(ART_E_INVALID, "art_scene_add_primitive: idx_bytes must be 2 or 4");
    if (n_indices == 0 || n_indices % 3 != 0) return fail(ART_E_INVALID, "art_scene_add_primitive: index count must be a positive multiple of 3");
    if (n_verts == 0 || tw == 0 || th == 0) return fail(ART_E_INVALID, "art_scene_add_primitive: empty vertices or texture");
    if (idx_bytes == 2 && n_verts > 65536) return fail(ART_E_INVALID, "art_scene_add_primitive: u16 indices cannot address the vertex count");
    for (int i = 0; i < 12; i++) if (!std::isfinite(model3x4[i])) return fail(ART_E_INVALID, "art_scene_add_primitive: non-finite model matrix");
    for (uint32_t i = 0; i < n_indices; i++) {
        uint32_t v = idx_bytes == 2 ? ((const uint16_t *)indices)[i] : ((const uint32_t *)indices)[i];
        if (v >= n_verts) return fail(ART_E_INVALID, "art_scene_add_primitive: index out of range");
    }
    HostPrim p;
    p.verts.assign(verts, verts + n_verts);
    p.indices.assign((const uint8_t *)indices, (const uint8_t *)indices + (size_t)n_indices * idx_bytes);
    p.n_indices = n_indices; p.idx_bytes = idx_bytes;
    p.tex.assign(rgba8, rgba8 + (size_t)3 * tw * th * 4);
    p.tw = tw; p.th = th;
    std::memcpy(p.o2w, model3x4, 48);
    affine_inverse(p.o2w, p.w2o);
    c->prims.push_back(std::move(p)); c->uploaded.clear();
    c->built = false;
    if (out_id) *out_id = (uint32_t)c->prims.size() - 1;
    return ART_OK;
}

int32_t art_scene_clear(ArtContext *c) {
    if (!c) return fail(ART_E_INVALID, "art_scene_clear: null context");
    c->prims.clear(); c->uploaded.clear(); c->built = false;
    return ART_OK;
}

int32_t art_scene_set_primitive_enabled(ArtContext *c, uint32_t id, int32_t enabled) {
    if (!c) return fail(ART_E_INVALID, "art_scene_set_primitive_enabled: null context");
    if (id >= c->prims.size()) return fail(ART_E_INVALID, "art_scene_set_primitive_enabled: no such primitive");
    HostPrim &p = c->prims[id];
    if (p.enabled == (enabled != 0)) return ART_OK;
    p.enabled = enabled != 0;
    if (!c->built) return ART_OK;                                // takes effect with the build
    if (p.n_indices < 3) return ART_OK;                          // no triangles: nothing to take out or bring back
    if (id < c->h_dev_prims.size() && c->h_dev_prims[id].n_tri > 0) {
        // Its triangles are in the built structure: they are masked (written "nowhere", every box above them shrunk) or restored by the refit in front of the
        // next frame, like a move -- a model that crosses the residency radius (vk_model.rs:334-345) costs a fraction of a millisecond, not a build; its
        // device arrays stay where they are until the next art_scene_build (288 GB of HBM: the way back is as cheap).
        DevPrim &d = c->h_dev_prims[id];
        d.masked = p.enabled ? 0u : 1u;
        c->masked_tris += p.enabled ? -(int64_t)d.n_tri : (int64_t)d.n_tri;
        c->prim_moved[id] = c->as_epoch + 1;
        c->xform_dirty = true;
        c->stats.num_triangles = (uint32_t)((int64_t)c->T - c->masked_tris);
        return ART_OK;
    }
    c->built = false;                                            // not part of the built structure: art_scene_build
    return ART_OK;
}

int32_t art_scene_needs_build(const ArtContext *c) {
    if (!c) return fail(ART_E_INVALID, "art_scene_needs_build: null context");
    return c->built ? 0 : 1;
}

int32_t art_scene_set_model_matrix(ArtContext *c, uint32_t first, uint32_t n, const float model3x4[12]) {
    if (!c || !model3x4) return fail(ART_E_INVALID, "art_scene_set_model_matrix: null argument");
    if (n == 0 || first >= c->prims.size() || n > c->prims.size() - first) return fail(ART_E_INVALID, "art_scene_set_model_matrix: no such primitives");
    for (int i = 0; i < 12; i++) if (!std::isfinite(model3x4[i])) return fail(ART_E_INVALID, "art_scene_set_model_matrix: non-finite matrix");
    float w2o[12];
    affine_inverse(model3x4, w2o);
    for (uint32_t id = first; id < first + n; id++) {
        HostPrim &p = c->prims[id];
        if (std::memcmp(p.o2w, model3x4, 48) == 0) continue;     // where it already is
        std::memcpy(p.o2w, model3x4, 48); std::memcpy(p.w2o, w2o, 48);
        if (!c->built || id >= c->h_dev_prims.size()) continue;  // takes effect with the build
        std::memcpy(c->h_dev_prims[id].o2w, model3x4, 48); std::memcpy(c->h_dev_prims[id].w2o, w2o, 48);
        if (id < c->prim_moved.size()) c->prim_moved[id] = c->as_epoch + 1;   // the next refit is the first to show it
        if (p.enabled) c->xform_dirty = true;                    // instanced: the next art_trace (or query) refits first
    }
    return ART_OK;
}

int32_t art_scene_build(ArtContext *c) {
    if (!c) return fail(ART_E_INVALID, "art_scene_build: null context");
    if (c->prims.empty()) return fail(ART_E_STATE, "art_scene_build: no primitives");
    int32_t r = use_device(c); if (r) return r;
    r = sync_all(c); if (r) return r;
    as_release(c); c->xform_dirty = false; c->as_epoch = 0; c->binary_epoch = 0; c->refit_cost_ratio = 1.0f; // the versions were copies of the tree that goes away
    lbvh_free(c->bvh); c->bvh.arena = &c->arena; c->built = false; drop_graphs(c);
    // Only enabled primitives are uploaded and instanced (get_acceleration_structure_instance returns None unless the model is in
    // the Device state, vk_model.rs:360-372).  Ids keep their meaning: a disabled primitive stays in the table with zero triangles.
    // With nothing enabled the tree is one zero-area triangle that no ray can hit (an empty TLAS: every ray misses).
    static const ArtVertex kNoVertex{};
    static const uint16_t kNoIndex[3] = {0, 0, 0};
    static const uint8_t kNoTexel[12] = {0};
    bool any = false;
    for (auto &p : c->prims) any = any || (p.enabled && p.n_indices >= 3);
    size_t nv = any ? 0 : 1, ib = any ? 0 : 16, nt = any ? 0 : 3; uint32_t T = 0;
    for (auto &p : c->prims) if (p.enabled) { nv += p.verts.size(); ib += (p.indices.size() + 15) & ~(size_t)15; nt += (size_t)3 * p.tw * p.th; }
    std::vector<uint8_t> now_set(c->prims.size());
    for (size_t k = 0; k < c->prims.size(); k++) now_set[k] = c->prims[k].enabled ? 1 : 0;
    const bool resident = any && now_set == c->uploaded;   // the same primitives as the last upload, nothing added since: their data is where this build would put it
    if (!resident) c->uploaded.clear();   // (an upload that fails half way leaves nothing to rely on)
    HIPC(c->d_verts.ensure(nv * 12)); HIPC(c->d_indices.ensure(ib)); HIPC(c->d_tex.ensure(nt));
    std::vector<DevPrim> dp(c->prims.size() + (any ? 0 : 1));
    std::vector<uint32_t> first(dp.size());
    size_t ov = 0, oi = 0, ot = 0;
    for (size_t k = 0; k < c->prims.size(); k++) {
        auto &p = c->prims[k];
        DevPrim &d = dp[k];
        std::memset(&d, 0, sizeof(d));
        d.vertices = c->d_verts.p + ov * 12; d.indices = c->d_indices.p + oi; d.texture_offset = (uint32_t)ot; d.single_index_size = p.idx_bytes;
        d.tw = p.tw; d.th = p.th; d.first_tri = T; d.n_tri = p.enabled ? p.n_indices / 3 : 0;
        std::memcpy(d.o2w, p.o2w, 48); std::memcpy(d.w2o, p.w2o, 48);
        first[k] = T;
        if (!p.enabled) continue;
        if (!resident) {
            HIPC(hipMemcpy(c->d_verts.p + ov * 12, p.verts.data(), p.verts.size() * 48, hipMemcpyHostToDevice));
            HIPC(hipMemcpy(c->d_indices.p + oi, p.indices.data(), p.indices.size(), hipMemcpyHostToDevice));
            HIPC(hipMemcpy(c->d_tex.p + ot, p.tex.data(), p.tex.size(), hipMemcpyHostToDevice));
        }
        T += d.n_tri;
        ov += p.verts.size(); oi += (p.indices.size() + 15) & ~(size_t)15; ot += (size_t)3 * p.tw * p.th;
    }
    if (!any) {
        DevPrim &d = dp.back();
        std::memset(&d, 0, sizeof(d));
        HIPC(hipMemcpy(c->d_verts.p, &kNoVertex, 48, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(c->d_indices.p, kNoIndex, 6, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(c->d_tex.p, kNoTexel, 12, hipMemcpyHostToDevice));
        d.vertices = c->d_verts.p; d.indices = c->d_indices.p; d.texture_offset = 0; d.single_index_size = 2; d.tw = 1; d.th = 1; d.first_tri = T; d.n_tri = 1;
        d.o2w[0] = d.o2w[5] = d.o2w[10] = 1.0f; d.w2o[0] = d.w2o[5] = d.w2o[10] = 1.0f;
        first.back() = T;
        T += 1;
    }
    HIPC(c->d_prims.ensure(dp.size())); HIPC(c->d_first_tri.ensure(first.size()));
    HIPC(hipMemcpy(c->d_prims.p, dp.data(), dp.size() * sizeof(DevPrim), hipMemcpyHostToDevice));
    HIPC(hipMemcpy(c->d_first_tri.p, first.data(), first.size() * 4, hipMemcpyHostToDevice));
    c->uploaded = any ? now_set : std::vector<uint8_t>();
    c->h_first_tri = first;
    c->h_dev_prims = dp; c->masked_tris = 0;
    c->prim_moved.assign(dp.size(), 0);
    c->T = T;
    BuildInputs in{c->d_prims.p, (uint32_t)dp.size(), c->d_first_tri.p, T, c->cfg.morton_bits};
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
    HIPC(hipEventRecord(e0, c->main_stream()));
    const bool own_tree = c->fast_trace && T >= 3;   // a PREFER_FAST_TRACE build makes its own tree over the leaves: the canonical tree's boxes are computed only when asked for (art_get_lbvh)
    hipError_t e = lbvh_build(in, c->bvh, c->main_stream(), !own_tree);
    c->bvh.log = c->tuning.log;
    if (e != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return hipfail(e, "lbvh_build"); }
    if (c->fast_trace) { // PREFER_FAST_TRACE (vk_model.rs:968): the traversal nodes get a SAH-driven topology over the same leaves
        bool done = false;
        if (c->tree_builder == 3) { // the binned SAH on the device
            e = sah_build_device(c->bvh, T, c->main_stream());
            if (e != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return hipfail(e, "sah_build_device"); }
            done = true;
        }
        if (!done) {
            e = sah_build(c->bvh, T, c->main_stream());
            if (e != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return hipfail(e, "sah_build"); }
        }
    }
    HIPC(hipEventRecord(e1, c->main_stream())); HIPC(hipEventSynchronize(e1));
    float ms = 0; HIPC(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    c->stats.build_ms = ms; c->stats.num_triangles = T; c->stats.num_primitives = (uint32_t)dp.size(); c->stats.num_nodes = c->kind_primary == 4 ? c->bvh.n_wide : (T > 1 ? T - 1 : 1);
    c->built = true;
    c->plan.next_sample = c->frame_no; c->plan.interval = 1; // a new scene: the heavy blocks are elsewhere
    c->first_move_ms = 0.f; c->versions_ms = 0.f;
    if (c->cfg.flags & ART_FLAG_DYNAMIC_SCENE) { r = as_create(c); if (r) return r; }   // the host said its models move: the ring of versions now, not in front of the first moved frame
    return ART_OK;
}

int32_t art_set_camera(ArtContext *c, const ArtCamera *cam) {
    if (!c || !cam) return fail(ART_E_INVALID, "art_set_camera: null argument");
    if (!camera_finite(cam)) return fail(ART_E_INVALID, "art_set_camera: non-finite value in the camera block");
    if (!c->have_camera || std::memcmp(&c->camera, cam, sizeof(ArtCamera)) != 0) drop_graphs(c); // the camera block is a kernel argument
    if (!c->have_camera || std::memcmp(&c->camera, cam, sizeof(ArtCamera)) != 0) plan_hint_moved(c); // the heavy blocks move with the view
    c->camera = *cam; c->have_camera = true;
    for (uint32_t i = 0; i + 1 < kMaxBatch; i++) c->cam_more[i] = *cam; // every frame of a launch, until art_set_camera_batch says otherwise
    return ART_OK;
}

int32_t art_set_frames_per_launch(ArtContext *c, uint32_t n) {
    if (!c || n == 0 || n > kMaxBatch) return fail(ART_E_INVALID, "art_set_frames_per_launch: 1..4");
    if (n > 1 && !(c->fused && c->kind_primary == 8 && c->kind_shadow == 8)) return fail(ART_E_STATE, "art_set_frames_per_launch: only the fused frame traces several frames per launch");
    if (n == c->B) return ART_OK;
    int32_t r = use_device(c); if (r) return r;
    r = sync_all(c); if (r) return r;
    drop_graphs(c);
    c->B = n; c->read_b = 0; c->frame_ready = false; c->traced = false; // the per-slot buffers are laid out again, n frames each
    for (uint32_t k = 0; k < c->F; k++) { c->slot[k].ext_tiles = nullptr; c->slot[k].ext_ring_n = 0; c->slot[k].tiles_of_last = nullptr; c->slot[k].ext_tiles_bytes = 0; }
    return ART_OK;
}
int32_t art_set_camera_batch(ArtContext *c, const ArtCamera *cams, uint32_t n) {
    if (!c || !cams) return fail(ART_E_INVALID, "art_set_camera_batch: null argument");
    if (n != c->B) return fail(ART_E_INVALID, "art_set_camera_batch: one camera per frame of a launch (art_set_frames_per_launch)");
    for (uint32_t i = 1; i < n; i++) if (!camera_finite(&cams[i])) return fail(ART_E_INVALID, "art_set_camera_batch: non-finite value in a camera block");
    int32_t r = art_set_camera(c, &cams[0]); if (r) return r;
    for (uint32_t i = 1; i < n; i++) c->cam_more[i - 1] = cams[i];
    return ART_OK;
}
int32_t art_set_read_frame(ArtContext *c, uint32_t b) {
    if (!c || b >= c->B) return fail(ART_E_INVALID, "art_set_read_frame: frame >= frames per launch");
    c->read_b = b;
    return ART_OK;
}
int32_t art_camera_from_params(const float pos[3], const float dir[3], float aspect, float fovy, float znear, float zfar, ArtCamera *out) {
    if (!pos || !dir || !out) return fail(ART_E_INVALID, "art_camera_from_params: null argument");
    // VkCamera::set_dir normalises (vk_camera.rs:133-136); view = look_at_rh(pos, pos + dir, up = (0,-1,0)) (:182-189)
    float dn[3], tgt[3], f[3], s[3], u[3];
    normalize3(dir, dn);
    for (int k = 0; k < 3; k++) tgt[k] = (pos[k] + dn[k]) - pos[k];
    normalize3(tgt, f);
    const float up[3] = {0.0f, -1.0f, 0.0f};
    float sx[3]; cross3h(f, up, sx); normalize3(sx, s);
    cross3h(s, f, u);
    float *V = out->view;
    V[0] = s[0]; V[4] = s[1]; V[8] = s[2]; V[12] = -dot3h(s, pos);
    V[1] = u[0]; V[5] = u[1]; V[9] = u[2]; V[13] = -dot3h(u, pos);
    V[2] = -f[0]; V[6] = -f[1]; V[10] = -f[2]; V[14] = dot3h(f, pos);
    V[3] = 0; V[7] = 0; V[11] = 0; V[15] = 1;
    // proj = Perspective3::new(aspect, fovy, znear, zfar) (:191-193): OpenGL convention, z in [-1, 1]
    float *P = out->proj;
    std::memset(P, 0, 64);
    float cc = 1.0f / std::tan(fovy * 0.5f);
    P[0] = cc / aspect; P[5] = cc; P[10] = (zfar + znear) / (znear - zfar); P[14] = 2.0f * zfar * znear / (znear - zfar); P[11] = -1.0f;
    // (a direction along the up axis (0, -1, 0) has no side vector: look_at_rh's cross product is zero and the view matrix NaN -- an error here, where the reference would render NaN)
    if (!mat4_inverse(out->view, out->view_inv) || !mat4_inverse(out->proj, out->proj_inv)) return fail(ART_E_INVALID, "art_camera_from_params: singular or non-finite view / projection matrix (direction along the up axis, zero field of view, znear == zfar ...)");
    out->camera_pos[0] = pos[0]; out->camera_pos[1] = pos[1]; out->camera_pos[2] = pos[2];
    return ART_OK;
}

int32_t art_set_lights(ArtContext *c, const ArtLight *lights, uint32_t n) {
    if (!c || (n && !lights)) return fail(ART_E_INVALID, "art_set_lights: null argument");
    if (n > kMaxLightsTotal) return fail(ART_E_INVALID, "art_set_lights: more than 1024 lights");
    for (uint32_t i = 0; i < n; i++) if (lights[i].type > 3) return fail(ART_E_INVALID, "art_set_lights: unknown light type");
    int32_t r = use_device(c); if (r) return r;
    bool resized = n != c->lights.size();
    bool same = !resized && (n == 0 || std::memcmp(c->lights.data(), lights, (size_t)n * sizeof(ArtLight)) == 0);
    if (same) return ART_OK; // like VkLights' dirty flag (vk_lights.rs:81-139)
    c->lights.assign(lights, lights + n); c->lights_epoch++;
    plan_hint_moved(c); // the shadow walks change: look at the waves again
    if (resized) { r = sync_all(c); if (r) return r; } // the per-frame buffers are resized with the light count
    drop_graphs(c); // the records are kernel arguments (FrameArgs::lights): nothing to upload, and frames in flight keep the ones they were launched with
    if (resized) c->frame_ready = false;
    return ART_OK;
}

static void zero_light(ArtLight *l) { std::memset(l, 0, sizeof(*l)); }
static void cp3(float *d, const float *s) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }

int32_t art_light_point(const float pos[3], const float color[3], float falloff, int32_t casts, ArtLight *o) { // lights.rs:144-159
    if (!pos || !color || !o) return fail(ART_E_INVALID, "art_light_point: null argument");
    zero_light(o); cp3(o->pos, pos); o->type = 0; o->casts_shadows = casts ? 1u : 0u; cp3(o->color, color); o->falloff_distance = falloff;
    return ART_OK;
}
int32_t art_light_spot(const float pos[3], const float dir[3], const float color[3], float falloff, float penumbra, float umbra, int32_t casts, ArtLight *o) { // lights.rs:228-243
    if (!pos || !dir || !color || !o) return fail(ART_E_INVALID, "art_light_spot: null argument");
    zero_light(o); cp3(o->pos, pos); o->type = 1; cp3(o->dir, dir); o->casts_shadows = casts ? 1u : 0u; cp3(o->color, color);
    o->falloff_distance = falloff; o->penumbra_angle = penumbra; o->umbra_angle = umbra;
    return ART_OK;
}
int32_t art_light_directional(const float dir[3], const float color[3], int32_t casts, ArtLight *o) { // lights.rs:281-296
    if (!dir || !color || !o) return fail(ART_E_INVALID, "art_light_directional: null argument");
    zero_light(o); o->type = 2; cp3(o->dir, dir); o->casts_shadows = casts ? 1u : 0u; cp3(o->color, color);
    return ART_OK;
}
int32_t art_light_area(const float pos[3], const float pos2[3], const float pos3[3], int32_t invert_normal, const float color[3], float falloff,
                       float penumbra, float umbra, int32_t casts, ArtLight *o) { // lights.rs:383-403
    if (!pos || !pos2 || !pos3 || !color || !o) return fail(ART_E_INVALID, "art_light_area: null argument");
    zero_light(o);
    float a[3] = {pos[0] - pos2[0], pos[1] - pos2[1], pos[2] - pos2[2]}, b[3] = {pos3[0] - pos2[0], pos3[1] - pos2[1], pos3[2] - pos2[2]}, n[3];
    cross3h(a, b, n);
    if (invert_normal) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    normalize3(n, n);
    cp3(o->pos, pos); o->type = 3; cp3(o->dir, n); o->casts_shadows = casts ? 1u : 0u; cp3(o->color, color); o->falloff_distance = falloff;
    cp3(o->area_pos2, pos2); o->penumbra_angle = penumbra; cp3(o->area_pos3, pos3); o->umbra_angle = umbra;
    return ART_OK;
}

int32_t art_resize(ArtContext *c, uint32_t w, uint32_t h) {
    if (!c || w == 0 || h == 0 || w > 16384 || h > 16384) return fail(ART_E_INVALID, "art_resize: bad extent");
    int32_t r = use_device(c); if (r) return r;
    r = sync_all(c); if (r) return r;
    c->W = w; c->H = h; c->frame_ready = false; c->traced = false;
    return ART_OK;
}

static FrameArgs make_frame_args(ArtContext *c, FrameSlot &S, uint32_t version) {
    FrameArgs a{};
    static_assert(sizeof(CameraArg) == sizeof(ArtCamera), "camera block layout");
    std::memcpy(&a.cam, &c->camera, sizeof(ArtCamera));
    a.W = c->W; a.H = c->H; a.tile_list = c->d_tile_list.p; a.n_tiles_owned = (uint32_t)c->tile_list.size(); a.tiles_x = c->tiles_x; a.n_local = c->n_local; a.block_order = c->d_block_order.p;
    const AsPtrs as = as_ptrs(c, version); // the version of the acceleration structure this launch reads
    a.nodes = c->bvh.nodes; a.wide = as.wide; a.widef = as.widef; a.packet_wide = c->packet_wide; a.trace_kind[0] = c->kind_primary; a.trace_kind[1] = c->kind_shadow; a.trace_kind[2] = c->kind_ao; a.tune = TraceTune{c->tuning.trace_chunk, c->tuning.trace_refill, c->tuning.trace_blocks, c->tuning.trace_leaf_batch}; a.pipelined = c->F > 1; a.tris = as.tris; a.shade_tris = c->bvh.shade_tris; a.prims = as.prims; a.tex_pool = c->d_tex.p;
    a.n_lights = (uint32_t)c->lights.size();
    const uint32_t n_arg = std::min(a.n_lights, (uint32_t)kMaxLights);
    if (n_arg) std::memcpy(a.lights, c->lights.data(), (size_t)n_arg * sizeof(ArtLight));
    for (uint32_t i = 0; i < n_arg; i++) directional_constants(a.lights[i]);
    a.lights_more = a.n_lights > (uint32_t)kMaxLights ? S.d_lights_more.p : nullptr;   // (art_trace brings the slot's table up to date: lights_upload)
    a.pix_more = (a.n_lights > (uint32_t)kMaxLights && c->fused) ? S.d_pix_more.p : nullptr;
    a.hits = S.d_hits.p; a.contrib = S.d_contrib.p; a.shadow_rays = S.d_shadow_rays.p; a.counters = S.d_counters.p;
    a.color = S.d_color.p; a.depth = S.d_depth.p; a.normal = S.d_normal.p;
    a.color_tiles = c->tiled() ? S.last_tiles() : nullptr; a.tiles_packed = c->tiles_packed(); // art_trace picks the frame's buffer (tiles_for)
    a.shadow_bits = (c->cfg.flags & ART_FLAG_KEEP_DEBUG) ? S.d_shadow_bits.p : nullptr;
    a.pix_bits = S.d_shadow_bits.p; a.keep_hits = (c->cfg.flags & ART_FLAG_KEEP_DEBUG) != 0;
    a.batch = c->B; a.tiles_stride = c->padded_tiles * kTilePixels;
    for (uint32_t i = 0; i + 1 < kMaxBatch; i++) std::memcpy(&a.cam_more[i], &c->cam_more[i], sizeof(ArtCamera));
    a.tile_xy = c->d_tile_xy.p; a.wave_items = c->plan.d_items[c->plan.cur].p; a.n_wave_items = c->plan.n_items[c->plan.cur]; a.wave_cost = nullptr; // art_trace sets it for the frames the wave plan samples
    return a;
}

// more than 16 lights: ring slot S's table of records 16.. as of the current list, on the slot's stream (in front of the frame that reads it)
static int32_t lights_upload(ArtContext *c, FrameSlot &S, hipStream_t s) {
    const size_t n = c->lights.size();
    if (n <= (size_t)kMaxLights || S.lights_epoch == c->lights_epoch) return ART_OK;
    const size_t m = n - kMaxLights;
    if (S.lights_ev_set) HIPC(hipEventSynchronize(S.lights_ev));   // the upload that read the staging copy last
    if (S.h_lights_cap < m) {
        if (S.h_lights_more) (void)hipHostFree(S.h_lights_more);
        S.h_lights_more = nullptr; S.h_lights_cap = 0;
        HIPC(hipHostMalloc((void **)&S.h_lights_more, m * sizeof(ArtLight), hipHostMallocDefault)); S.h_lights_cap = m;
    }
    if (S.d_lights_more.n < m) { HIPC(hipStreamSynchronize(s)); HIPC(S.d_lights_more.ensure(m)); }   // (frames of this slot still read the old table)
    std::memcpy(S.h_lights_more, c->lights.data() + kMaxLights, m * sizeof(ArtLight));
    for (size_t i = 0; i < m; i++) directional_constants(S.h_lights_more[i]);
    HIPC(hipMemcpyAsync(S.d_lights_more.p, S.h_lights_more, m * sizeof(ArtLight), hipMemcpyHostToDevice, s));
    if (!S.lights_ev) HIPC(hipEventCreateWithFlags(&S.lights_ev, hipEventDisableTiming));
    HIPC(hipEventRecord(S.lights_ev, s)); S.lights_ev_set = true;
    S.lights_epoch = c->lights_epoch;
    return ART_OK;
}

int32_t art_trace(ArtContext *c) {
    if (!c) return fail(ART_E_INVALID, "art_trace: null context");
    if (!c->built) return fail(ART_E_STATE, "art_trace: scene not built (art_scene_build)");
    if (!c->have_camera) return fail(ART_E_STATE, "art_trace: no camera (art_set_camera)");
    if (c->W == 0 || c->H == 0) return fail(ART_E_STATE, "art_trace: zero extent (art_resize)");
    int32_t r = use_device(c); if (r) return r;
    if (!c->frame_ready) { r = sync_all(c); if (r) return r; r = setup_frame(c); if (r) return r; }
    const uint32_t k = (uint32_t)(c->frame_no % c->F);
    FrameSlot &S = c->slot[k];
    hipStream_t s = c->stream_of(k);
    if (S.wait_event) { HIPC(hipStreamWaitEvent(s, (hipEvent_t)S.wait_event, 0)); S.wait_event = nullptr; }
    if (c->xform_dirty) { r = scene_refresh(c, k, s); if (r) return r; } // a model moved: the refit this frame is ordered behind (past the cost threshold: a rebuild)
    r = ensure_wide(c, c->kind_primary == 4 || c->kind_shadow == 4 || c->packet_wide); if (r) return r;
    r = ensure_binary(c, !c->packet_wide || c->kind_primary == 2 || c->kind_shadow == 2); if (r) return r;
    if (c->plan.enabled) { r = plan_poll(c); if (r) return r; }
    const uint32_t ver = c->as_cur;
    if (!c->as.empty()) {
        AsVersion &V = c->as[ver];
        if (!V.ready_known) { // the refit that wrote this version may still run on another ring slot's stream
            if (hipEventQuery(V.ready) == hipSuccess) V.ready_known = true;
            else if (V.ready_slot != k) HIPC(hipStreamWaitEvent(s, V.ready, 0));
        }
        V.used[k] = c->frame_no + 1; V.aux[k] = false;
    }
    S.as_version = ver;
    r = lights_upload(c, S, s); if (r) return r;
    FrameArgs a = make_frame_args(c, S, ver);
    if (c->tiled()) a.color_tiles = S.tiles_for(c->frame_no, c->F); // alternates when a pair of buffers is bound
    const bool fused = c->fused && c->kind_primary == 8 && c->kind_shadow == 8;
    if (c->B > 1 && !fused) return fail(ART_E_STATE, "art_trace: several frames per launch need the default fused frame");
    hipEvent_t *ev = c->ev[c->frame_no % ArtContext::kRing];
    c->ev_fused[c->frame_no % ArtContext::kRing] = fused;
    if (c->graph_mode && !fused && S.ext_ring_n < 2) { // a fused frame is a single launch: nothing for a graph to save; alternating tile buffers change a kernel argument
        if (!S.graph) { // capture the frame once per slot; stage events are not part of it
            hipGraph_t g = nullptr;
            HIPC(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            hipError_t e = fused ? hipSuccess : hipMemsetAsync(S.d_counters.p, 0, kCounterWords * 4, s);
            if (e == hipSuccess && a.n_local) { if (fused) launch_frame(a, s); else { launch_primary(a, s); launch_shade(a, s); launch_shadow(a, s); launch_accumulate(a, s); } e = hipGetLastError(); }
            hipError_t e2 = hipStreamEndCapture(s, &g);
            if (e != hipSuccess || e2 != hipSuccess) { if (g) (void)hipGraphDestroy(g); return hipfail(e != hipSuccess ? e : e2, "art_trace: graph capture"); }
            e = hipGraphInstantiate(&S.graph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) { S.graph = nullptr; return hipfail(e, "hipGraphInstantiate"); }
        }
        for (int i = 0; i < 4; i++) HIPC(hipEventRecord(ev[i], s));
        HIPC(hipGraphLaunch(S.graph, s));
        HIPC(hipEventRecord(ev[4], s));
        HIPC(hipEventRecord(S.done, s)); S.done_alias = nullptr;
        S.ao_valid = false; S.presented = false;
        c->last = k; c->frame_no++; c->traced = true;
        return ART_OK;
    }
    if (fused) { // one launch; its time is booked on the first stage
        HIPC(hipEventRecord(ev[0], s));
        WavePlan &P = c->plan;
        const bool sample = ((P.enabled && c->frame_no >= P.next_sample) || c->force_sample) && !P.pending && a.n_wave_items && plan_other_free(c);
        if (sample) a.wave_cost = S.d_wave_cost.p;
        const bool counted = a.n_local ? launch_frame(a, s) : false;
        HIPC(hipEventRecord(ev[4], s));
        S.done_alias = ev[4];           // also the frame's completion event (a record is a packet in the frame's queue: 1/8 share 33 -> 29 us)
        HIPC(hipGetLastError());
        if (counted) { r = plan_launch(c, a, ev[4]); if (r) return r; } // now and then a frame counts its waves' packet steps: the next plan is made from them, behind the frame, on the device
        S.ao_valid = false; S.presented = false;
        c->last = k; c->frame_no++; c->traced = true;
        return ART_OK;
    }
    S.done_alias = nullptr;
    HIPC(hipMemsetAsync(S.d_counters.p, 0, kCounterWords * 4, s));   // the staged frame's work cursors and count slots
    HIPC(hipEventRecord(ev[0], s));
    if (a.n_local) launch_primary(a, s);
    HIPC(hipEventRecord(ev[1], s));
    if (a.n_local) launch_shade(a, s);
    HIPC(hipEventRecord(ev[2], s));
    if (a.n_local) launch_shadow(a, s);
    HIPC(hipEventRecord(ev[3], s));
    if (a.n_local) launch_accumulate(a, s);
    HIPC(hipEventRecord(ev[4], s));
    HIPC(hipEventRecord(S.done, s));
    HIPC(hipGetLastError());
    S.ao_valid = false; S.presented = false;
    c->last = k;
    c->frame_no++;
    c->traced = true;
    return ART_OK;
}

int32_t art_sample_wave_steps(ArtContext *c, uint32_t *items, uint32_t *steps, uint32_t cap, uint32_t *n) {
    if (!c || !n) return fail(ART_E_INVALID, "art_sample_wave_steps: null argument");
    int32_t r = use_device(c); if (r) return r;
    r = sync_all(c); if (r) return r;
    if (c->frame_ready) { r = plan_poll(c); if (r) return r; }   // a plan that has landed takes effect first
    if (c->plan.pending) return fail(ART_E_STATE, "art_sample_wave_steps: a sample is still in flight");
    c->force_sample = true;
    r = art_trace(c);
    c->force_sample = false;
    if (r) return r;
    r = sync_all(c); if (r) return r;
    WavePlan &P = c->plan;
    if (!P.pending) return fail(ART_E_STATE, "art_sample_wave_steps: this context's frames are not the fused frame (no step counts)");
    *n = P.n_items[P.pending_table];
    const uint32_t m = std::min(*n, cap);
    if (items && m) HIPC(hipMemcpy(items, P.d_items[P.pending_table].p, (size_t)m * 8, hipMemcpyDeviceToHost));
    if (steps && m) HIPC(hipMemcpy(steps, c->slot[c->last].d_wave_cost.p, (size_t)m * 4, hipMemcpyDeviceToHost));
    if (!P.enabled) P.pending = false;   // nobody polls a plan that is switched off (k_plan ran all the same: its table is not adopted)
    return ART_OK;
}

int32_t art_sync(ArtContext *c) {
    if (!c) return fail(ART_E_INVALID, "art_sync: null context");
    int32_t r = use_device(c); if (r) return r;
    return sync_all(c);
}

int32_t art_trace_ao(ArtContext *c, uint32_t spp, float radius) {
    if (!c) return fail(ART_E_INVALID, "art_trace_ao: null context");
    if (!c->traced || !c->frame_ready) return fail(ART_E_STATE, "art_trace_ao: call art_trace first (AO consumes that frame's depth + normal outputs)");
    if (spp == 0 || spp > 64 || !(radius > 0.0f)) return fail(ART_E_INVALID, "art_trace_ao: spp must be 1..64 and radius > 0");
    if (c->B > 1) return fail(ART_E_STATE, "art_trace_ao: not with several frames per launch");
    int32_t r = use_device(c); if (r) return r;
    r = ensure_wide(c, c->kind_ao == 4 || c->kind_ao == 6); if (r) return r;
    FrameSlot &S = c->slot[c->last];
    hipStream_t s = c->stream_of(c->last);
    const size_t n_occl = (size_t)c->n_local * (((spp + 3u) >> 2) * 4u); // one byte per slot of the per-ray tracer: groups of four samples (art_trace.hip ao_slot_decode)
    if (S.d_occl.n < n_occl || S.d_ao.n < (size_t)c->W * c->H || S.d_ao_pix.n < 2 * (size_t)c->n_local) {
        HIPC(hipStreamSynchronize(s));
        HIPC(S.d_occl.ensure(n_occl)); HIPC(S.d_ao.ensure((size_t)c->W * c->H)); HIPC(S.d_ao_pix.ensure(2 * (size_t)c->n_local));
        HIPC(hipMemset(S.d_ao.p, 0, (size_t)c->W * c->H * 4)); HIPC(hipDeviceSynchronize());
    }
    if (c->ao_tab_spp != spp) { // the sample directions in the tangent frame: a function of (sample, position in the 64x64 noise tile) only
        r = sync_all(c); if (r) return r;
        HIPC(c->d_ao_tab.ensure((size_t)spp * kAoTableEntriesPerSample));
        launch_ao_table(spp, c->d_ao_tab.p, c->main_stream());
        HIPC(hipGetLastError()); HIPC(hipStreamSynchronize(c->main_stream()));
        c->ao_tab_spp = spp;
    }
    uint32_t lut[65] = {0};
    for (uint32_t k = 0; k <= spp; k++) lut[k] = (uint32_t)(std::pow(1.0 - (double)k / (double)spp, 2.2) * 255.0 + 0.5); // XE_GTAO_DEFAULT_FINAL_VALUE_POWER (vk_xe_gtao.rs:22)
    r = ensure_binary(c, c->kind_ao == 2); if (r) return r;
    FrameArgs a = make_frame_args(c, S, S.as_version); // the structure the frame itself was traced in
    if (!c->as.empty()) c->as[S.as_version].aux[c->last] = true;
    HIPC(hipMemsetAsync(S.d_counters.p + 64 + 16 * 32, 0, 8 * 32 * 4, s)); // the AO launch's work cursors
    HIPC(hipEventRecord(S.ao_ev[0], s));
    if (a.n_local) launch_ao(a, spp, radius, S.d_occl.p, S.d_ao_pix.p, c->d_ao_tab.p, c->ao_entry, S.d_ao.p, lut, s);
    HIPC(hipEventRecord(S.ao_ev[1], s));
    HIPC(hipEventRecord(S.done, s)); S.done_alias = nullptr;
    HIPC(hipGetLastError());
    c->stats.ao_rays = 0; c->ao_spp = spp; S.ao_valid = true;
    return ART_OK;
}

int32_t art_present(ArtContext *c) {
    if (!c) return fail(ART_E_INVALID, "art_present: null context");
    if (!c->traced || !c->frame_ready) return fail(ART_E_STATE, "art_present: call art_trace first");
    if (c->B > 1) return fail(ART_E_STATE, "art_present: not with several frames per launch");
    int32_t r = use_device(c); if (r) return r;
    FrameSlot &S = c->slot[c->last];
    hipStream_t s = c->stream_of(c->last);
    size_t npix = (size_t)c->W * c->H;
    if (S.d_bgra.n < npix) { HIPC(hipStreamSynchronize(s)); HIPC(S.d_pcolor.ensure(npix)); HIPC(S.d_pnormal.ensure(npix)); HIPC(S.d_bgra.ensure(npix)); HIPC(S.d_pdepth.ensure(npix)); }
    uint32_t ctl[96];
    const float sat[3] = {0.0f, 0.0f, 0.0f}, ct[3] = {1.0f, 0.5f, 1.0f / 32.0f};
    lpm_control_block(false, 0.0f, 256.0f, 8.0f, 0.25f, 1.0f, sat, ct, ctl); // the parameters of vk_tonemap.rs:417-426
    launch_present((uint32_t)npix, S.d_color.p, S.d_normal.p, S.d_depth.p, S.ao_valid ? S.d_ao.p : nullptr, ctl, S.d_pcolor.p, S.d_pnormal.p, S.d_pdepth.p, S.d_bgra.p, s);
    HIPC(hipEventRecord(S.done, s)); S.done_alias = nullptr;
    HIPC(hipGetLastError());
    S.presented = true;
    return ART_OK;
}
int32_t art_lpm_control_block(int32_t shoulder, float soft_gap, float hdr_max, float exposure, float contrast, float shoulder_contrast, const float saturation[3],
                              const float crosstalk[3], uint32_t ctl[96]) {
    if (!saturation || !crosstalk || !ctl) return fail(ART_E_INVALID, "art_lpm_control_block: null argument");
    lpm_control_block(shoulder != 0, soft_gap, hdr_max, exposure, contrast, shoulder_contrast, saturation, crosstalk, ctl);
    return ART_OK;
}

static int32_t read_back(ArtContext *c, const void *src, size_t have, void *dst, size_t bytes, const char *who) {
    if (!c || !dst) return fail(ART_E_INVALID, std::string(who) + ": null argument");
    if (!c->traced) return fail(ART_E_STATE, std::string(who) + ": nothing traced yet");
    if (bytes != have) return fail(ART_E_INVALID, std::string(who) + ": size mismatch");
    int32_t r = use_device(c); if (r) return r;
    HIPC(hipStreamSynchronize(c->stream_of(c->last)));
    HIPC(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return ART_OK;
}
int32_t art_read_color(ArtContext *c, void *dst, size_t bytes) { return read_back(c, c ? c->slot[c->last].d_color.p + (size_t)c->read_b * c->W * c->H : nullptr, c ? (size_t)c->W * c->H * 16 : 0, dst, bytes, "art_read_color"); }
int32_t art_read_depth(ArtContext *c, void *dst, size_t bytes) { return read_back(c, c ? c->slot[c->last].d_depth.p + (size_t)c->read_b * c->W * c->H : nullptr, c ? (size_t)c->W * c->H * 4 : 0, dst, bytes, "art_read_depth"); }
int32_t art_read_normal(ArtContext *c, void *dst, size_t bytes) { return read_back(c, c ? c->slot[c->last].d_normal.p + (size_t)c->read_b * c->W * c->H : nullptr, c ? (size_t)c->W * c->H * 16 : 0, dst, bytes, "art_read_normal"); }

static int32_t dev_ptr(ArtContext *c, void *p, size_t n, void **out, size_t *bytes, const char *who) {
    if (!c || !out) return fail(ART_E_INVALID, std::string(who) + ": null argument");
    if (!c->frame_ready) { int32_t r = use_device(c); if (r) return r; if (c->W == 0 || c->H == 0) return fail(ART_E_STATE, std::string(who) + ": zero extent"); r = setup_frame(c); if (r) return r; }
    (void)p;
    *out = nullptr; if (bytes) *bytes = n;
    return ART_OK;
}
int32_t art_read_ao(ArtContext *c, void *dst, size_t bytes) {
    if (c && c->ao_spp == 0) return fail(ART_E_STATE, "art_read_ao: art_trace_ao has not run");
    return read_back(c, c ? c->slot[c->last].d_ao.p : nullptr, c ? (size_t)c->W * c->H * 4 : 0, dst, bytes, "art_read_ao");
}
static int32_t need_present(ArtContext *c, const char *who) {
    if (c && !c->slot[c->last].presented) return fail(ART_E_STATE, std::string(who) + ": art_present has not run for the latest frame");
    return ART_OK;
}
int32_t art_read_present(ArtContext *c, void *dst, size_t bytes) {
    int32_t r = need_present(c, "art_read_present"); if (r) return r;
    return read_back(c, c ? c->slot[c->last].d_bgra.p : nullptr, c ? (size_t)c->W * c->H * 4 : 0, dst, bytes, "art_read_present");
}
int32_t art_read_packed(ArtContext *c, void *color_b10g11r11, void *normal_b10g11r11, void *depth_f16) {
    int32_t r = need_present(c, "art_read_packed"); if (r) return r;
    if (!c) return fail(ART_E_INVALID, "art_read_packed: null context");
    size_t npix = (size_t)c->W * c->H;
    if (color_b10g11r11) { r = read_back(c, c->slot[c->last].d_pcolor.p, npix * 4, color_b10g11r11, npix * 4, "art_read_packed"); if (r) return r; }
    if (normal_b10g11r11) { r = read_back(c, c->slot[c->last].d_pnormal.p, npix * 4, normal_b10g11r11, npix * 4, "art_read_packed"); if (r) return r; }
    if (depth_f16) { r = read_back(c, c->slot[c->last].d_pdepth.p, npix * 2, depth_f16, npix * 2, "art_read_packed"); if (r) return r; }
    return ART_OK;
}
int32_t art_device_color(ArtContext *c, void **p, size_t *b) { int32_t r = dev_ptr(c, nullptr, 0, p, b, "art_device_color"); if (r) return r; *p = c->slot[c->last].d_color.p + (size_t)c->read_b * c->W * c->H; if (b) *b = (size_t)c->W * c->H * 16; return ART_OK; }
int32_t art_device_depth(ArtContext *c, void **p, size_t *b) { int32_t r = dev_ptr(c, nullptr, 0, p, b, "art_device_depth"); if (r) return r; *p = c->slot[c->last].d_depth.p + (size_t)c->read_b * c->W * c->H; if (b) *b = (size_t)c->W * c->H * 4; return ART_OK; }
int32_t art_device_normal(ArtContext *c, void **p, size_t *b) { int32_t r = dev_ptr(c, nullptr, 0, p, b, "art_device_normal"); if (r) return r; *p = c->slot[c->last].d_normal.p + (size_t)c->read_b * c->W * c->H; if (b) *b = (size_t)c->W * c->H * 16; return ART_OK; }

int32_t art_shard_layout(uint32_t width, uint32_t height, uint32_t shard_count, uint32_t shard_rank, uint32_t root_relief, uint32_t *tiles, uint32_t cap, uint32_t *owned, uint32_t *padded) {
    if (width == 0 || height == 0) return fail(ART_E_INVALID, "art_shard_layout: zero extent");
    if (root_relief > 255) return fail(ART_E_INVALID, "art_shard_layout: root_relief 0..255");
    uint32_t count = shard_count > 1 ? shard_count : 1;
    if (shard_rank >= count) return fail(ART_E_INVALID, "art_shard_layout: shard_rank >= shard_count");
    uint32_t tx_n = (width + kTile - 1) / kTile, ty_n = (height + kTile - 1) / kTile, mine = 0;
    if (count > 255) return fail(ART_E_INVALID, "art_shard_layout: at most 255 shards");
    std::vector<uint32_t> per(count, 0);
    const std::vector<uint8_t> owner_of = shard_owner_table(tx_n, ty_n, count, root_relief);
    for (uint32_t ty = 0; ty < ty_n; ty++)
        for (uint32_t tx = 0; tx < tx_n; tx++) {
            uint32_t o = owner_of[(size_t)ty * tx_n + tx];
            per[o]++;
            if (o == shard_rank) { if (tiles && mine < cap) tiles[mine] = ty * tx_n + tx; mine++; }
        }
    if (tiles && mine > cap) return fail(ART_E_INVALID, "art_shard_layout: tiles buffer too small");
    uint32_t mx = 0; for (uint32_t v : per) mx = v > mx ? v : mx;
    if (owned) *owned = mine;
    if (padded) *padded = mx;
    return ART_OK;
}
int32_t art_shard_tile_count(ArtContext *c, uint32_t *owned, uint32_t *padded) {
    if (!c) return fail(ART_E_INVALID, "art_shard_tile_count: null context");
    void *p; int32_t r = dev_ptr(c, nullptr, 0, &p, nullptr, "art_shard_tile_count"); if (r) return r;
    if (owned) *owned = (uint32_t)c->tile_list.size();
    if (padded) *padded = c->padded_tiles;
    return ART_OK;
}
int32_t art_device_color_tiles(ArtContext *c, void **p, size_t *b) {
    int32_t r = dev_ptr(c, nullptr, 0, p, b, "art_device_color_tiles"); if (r) return r;
    if (!c->tiled()) return fail(ART_E_STATE, "art_device_color_tiles: context is not sharded");
    FrameSlot &S = c->slot[c->last];
    const size_t one = (size_t)c->padded_tiles * kTilePixels * c->tile_px_bytes();
    *p = (char *)S.last_tiles() + c->read_b * one; if (b) *b = one;
    return ART_OK;
}
int32_t art_bind_color_tiles(ArtContext *c, uint32_t slot, void *dev, size_t bytes) {
    if (!c) return fail(ART_E_INVALID, "art_bind_color_tiles: null context");
    if (!c->tiled()) return fail(ART_E_STATE, "art_bind_color_tiles: context is not sharded");
    if (slot >= c->F) return fail(ART_E_INVALID, "art_bind_color_tiles: slot >= frames in flight");
    void *p; int32_t r = dev_ptr(c, nullptr, 0, &p, nullptr, "art_bind_color_tiles"); if (r) return r;
    if (dev && bytes != (size_t)c->padded_tiles * kTilePixels * c->tile_px_bytes() * c->B) return fail(ART_E_INVALID, "art_bind_color_tiles: size mismatch (padded tiles x tile bytes x frames per launch)");
    HIPC(hipStreamSynchronize(c->stream_of(slot)));
    c->slot[slot].ext_tiles = (float4 *)dev; c->slot[slot].ext_ring_n = 0; c->slot[slot].tiles_of_last = nullptr; c->slot[slot].ext_tiles_bytes = dev ? bytes : 0;
    drop_graphs(c);
    return ART_OK;
}
int32_t art_bind_color_tiles_pair(ArtContext *c, uint32_t slot, void *dev_even, void *dev_odd, size_t bytes) {
    if (!dev_even || !dev_odd) return fail(ART_E_INVALID, "art_bind_color_tiles_pair: null buffer");
    void *two[2] = {dev_even, dev_odd};
    return art_bind_color_tiles_ring(c, slot, two, 2, bytes);
}
int32_t art_bind_color_tiles_ring(ArtContext *c, uint32_t slot, void *const *bufs, uint32_t n, size_t bytes) {
    if (!bufs || n == 0 || n > FrameSlot::kTileRing) return fail(ART_E_INVALID, "art_bind_color_tiles_ring: 1..8 buffers");
    for (uint32_t i = 0; i < n; i++) if (!bufs[i]) return fail(ART_E_INVALID, "art_bind_color_tiles_ring: null buffer");
    int32_t r = art_bind_color_tiles(c, slot, bufs[0], bytes); if (r) return r;
    for (uint32_t i = 0; i < n; i++) c->slot[slot].ext_ring[i] = (float4 *)bufs[i];
    c->slot[slot].ext_ring_n = n;
    return ART_OK;
}
int32_t art_set_graph_mode(ArtContext *c, int32_t on) {
    if (!c) return fail(ART_E_INVALID, "art_set_graph_mode: null context");
    if (on && c->ext_stream) return fail(ART_E_STATE, "art_set_graph_mode: not with an external stream");
    c->graph_mode = on != 0;
    if (!on) drop_graphs(c);
    return ART_OK;
}
int32_t art_frames_in_flight(ArtContext *c, uint32_t *frames, uint32_t *next_slot) {
    if (!c) return fail(ART_E_INVALID, "art_frames_in_flight: null context");
    if (frames) *frames = c->F;
    if (next_slot) *next_slot = (uint32_t)(c->frame_no % c->F);
    return ART_OK;
}
int32_t art_frames_done(ArtContext *c, uint64_t first, uint32_t count, int32_t *done, uint64_t *traced) {
    if (!c || !done) return fail(ART_E_INVALID, "art_frames_done: null argument");
    if (traced) *traced = c->frame_no;
    *done = 0;
    if (count == 0) { *done = 1; return ART_OK; }
    if (first + count > c->frame_no) return fail(ART_E_INVALID, "art_frames_done: frames not traced yet");
    if (c->frame_no - first > (uint64_t)ArtContext::kRing) return fail(ART_E_INVALID, "art_frames_done: older than the 128 frames whose events are kept");
    int32_t r = use_device(c); if (r) return r;
    for (uint64_t f = first + count; f-- > first;) { // the newest first: it is the likeliest to be still running
        hipError_t e = hipEventQuery(c->ev[f % ArtContext::kRing][4]);
        if (e == hipErrorNotReady) return ART_OK;
        if (e != hipSuccess) return hipfail(e, "art_frames_done: hipEventQuery");
    }
    *done = 1;
    return ART_OK;
}
int32_t art_stream_wait_frame(ArtContext *c, void *hip_stream) {
    if (!c) return fail(ART_E_INVALID, "art_stream_wait_frame: null context");
    if (!c->traced) return fail(ART_E_STATE, "art_stream_wait_frame: nothing traced yet");
    int32_t r = use_device(c); if (r) return r;
    { FrameSlot &S = c->slot[c->last]; HIPC(hipStreamWaitEvent((hipStream_t)hip_stream, S.done_alias ? S.done_alias : S.done, 0)); }
    return ART_OK;
}
int32_t art_trace_for_stream(ArtContext *c, void *hip_stream, uint32_t *slot_used) {
    if (!c) return fail(ART_E_INVALID, "art_trace_for_stream: null context");
    const uint32_t k = (uint32_t)(c->frame_no % c->F);
    int32_t r = art_trace(c); if (r) return r;
    if (slot_used) *slot_used = k;
    return art_stream_wait_frame(c, hip_stream);
}
int32_t art_wait_external_event(ArtContext *c, void *hip_event) {
    if (!c) return fail(ART_E_INVALID, "art_wait_external_event: null context");
    c->slot[c->frame_no % c->F].wait_event = hip_event;
    return ART_OK;
}
int32_t art_collect_timings(ArtContext *c, float sums_ms[5], uint32_t *n_frames) {
    if (!c || !sums_ms || !n_frames) return fail(ART_E_INVALID, "art_collect_timings: null argument");
    int32_t r = use_device(c); if (r) return r;
    r = sync_all(c); if (r) return r;
    uint64_t from = c->collected_upto;
    if (c->frame_no - from > (uint64_t)ArtContext::kRing) from = c->frame_no - ArtContext::kRing;
    for (int k = 0; k < 5; k++) sums_ms[k] = 0.f;
    for (uint64_t f = from; f < c->frame_no; f++) {
        hipEvent_t *ev = c->ev[f % ArtContext::kRing];
        float ms = 0;
        if (c->ev_fused[f % ArtContext::kRing]) { HIPC(hipEventElapsedTime(&ms, ev[0], ev[4])); sums_ms[0] += ms; sums_ms[4] += ms; continue; } // one launch: booked on the first stage
        for (int k = 0; k < 4; k++) { HIPC(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); sums_ms[k] += ms; }
        HIPC(hipEventElapsedTime(&ms, ev[0], ev[4])); sums_ms[4] += ms;
    }
    *n_frames = (uint32_t)(c->frame_no - from);
    c->collected_upto = c->frame_no;
    return ART_OK;
}
int32_t art_read_color_tiles(ArtContext *c, void *dst, size_t bytes) {
    if (c && !c->tiled()) return fail(ART_E_STATE, "art_read_color_tiles: context is not sharded");
    FrameSlot *S = c ? &c->slot[c->last] : nullptr;
    const size_t one = c ? (size_t)c->padded_tiles * kTilePixels * c->tile_px_bytes() : 0;
    return read_back(c, S ? (const char *)S->last_tiles() + c->read_b * one : nullptr, one, dst, bytes, "art_read_color_tiles");
}
int32_t art_untile_gathered_frames(ArtContext *c, const void *gathered_dev, uint32_t shard_count, uint32_t shard_stride_tiles, uint32_t n_frames, void *frames_dev, void *hip_stream) {
    if (!c || !gathered_dev) return fail(ART_E_INVALID, "art_untile_gathered: null argument");
    if (shard_stride_tiles < c->padded_tiles) return fail(ART_E_INVALID, "art_untile_gathered: stride smaller than a shard's padded tile count");
    if (n_frames == 0 || (n_frames > 1 && (!frames_dev || (uint64_t)n_frames * c->padded_tiles > shard_stride_tiles))) return fail(ART_E_INVALID, "art_untile_gathered_frames: frames do not fit the shard stride (or no output given)");
    void *p; int32_t r = dev_ptr(c, nullptr, 0, &p, nullptr, "art_untile_gathered"); if (r) return r;
    if (shard_count != (c->cfg.shard_count > 1 ? c->cfg.shard_count : 1)) return fail(ART_E_INVALID, "art_untile_gathered: shard_count differs from the context's");
    r = use_device(c); if (r) return r;
    hipStream_t us = hip_stream ? (hipStream_t)hip_stream : c->stream_of(c->last);
    if (c->tiles_packed()) { // the gathered tiles are B10G11R11 words: the frame is the packed colour image (art_read_packed)
        FrameSlot &S = c->slot[c->last];
        if (!frames_dev && S.d_pcolor.n < (size_t)c->W * c->H) { HIPC(hipStreamSynchronize(us)); HIPC(S.d_pcolor.ensure((size_t)c->W * c->H)); }
        launch_untile_packed((const uint32_t *)gathered_dev, c->d_tile_slot.p, shard_stride_tiles, n_frames, c->padded_tiles, c->W, c->H, frames_dev ? (uint32_t *)frames_dev : S.d_pcolor.p, us);
    } else
    launch_untile((const float4 *)gathered_dev, c->d_tile_slot.p, shard_stride_tiles, n_frames, c->padded_tiles, c->W, c->H, frames_dev ? (float4 *)frames_dev : c->slot[c->last].d_color.p, us);
    HIPC(hipGetLastError());
    c->traced = true;
    return ART_OK;
}
int32_t art_untile_gathered_strided(ArtContext *c, const void *gathered_dev, uint32_t shard_count, uint32_t shard_stride_tiles, void *frame_dev, void *hip_stream) {
    return art_untile_gathered_frames(c, gathered_dev, shard_count, shard_stride_tiles, 1, frame_dev, hip_stream);
}
int32_t art_untile_gathered(ArtContext *c, const void *gathered_dev, uint32_t shard_count, void *frame_dev, void *hip_stream) {
    if (!c) return fail(ART_E_INVALID, "art_untile_gathered: null argument");
    return art_untile_gathered_strided(c, gathered_dev, shard_count, c->padded_tiles, frame_dev, hip_stream);
}

int32_t art_get_layout(ArtContext *c, ArtLayout *out) {
    if (!c || !out) return fail(ART_E_INVALID, "art_get_layout: null argument");
    void *p; int32_t r = dev_ptr(c, nullptr, 0, &p, nullptr, "art_get_layout"); if (r) return r;   // lays the frame out if that has not happened yet
    std::memset(out, 0, sizeof(*out));
    out->width = c->W; out->height = c->H; out->frames_in_flight = c->F; out->frames_per_launch = c->B;
    out->shard_rank = c->cfg.shard_count > 1 ? c->cfg.shard_rank : 0; out->shard_count = c->cfg.shard_count > 1 ? c->cfg.shard_count : 1;
    out->tiles_owned = (uint32_t)c->tile_list.size(); out->tiles_padded = c->padded_tiles; out->tile_bytes = c->tiled() ? (uint32_t)(kTilePixels * c->tile_px_bytes()) : 0u; // 0: no compact tile buffer
    return ART_OK;
}
int32_t art_timestamp_mark(ArtContext *c, uint32_t which) {
    if (!c || which > 1) return fail(ART_E_INVALID, "art_timestamp_mark: mark 0 or 1");
    if (!c->traced) return fail(ART_E_STATE, "art_timestamp_mark: nothing traced yet");
    int32_t r = use_device(c); if (r) return r;
    if (!c->mark[which]) HIPC(hipEventCreate(&c->mark[which]));
    HIPC(hipEventRecord(c->mark[which], c->stream_of(c->last)));
    return ART_OK;
}
int32_t art_timestamp_elapsed(ArtContext *c, float *ms) {
    if (!c || !ms) return fail(ART_E_INVALID, "art_timestamp_elapsed: null argument");
    if (!c->mark[0] || !c->mark[1]) return fail(ART_E_STATE, "art_timestamp_elapsed: both marks must have been recorded");
    int32_t r = use_device(c); if (r) return r;
    HIPC(hipEventSynchronize(c->mark[1]));
    HIPC(hipEventElapsedTime(ms, c->mark[0], c->mark[1]));
    return ART_OK;
}

int32_t art_get_stats(ArtContext *c, ArtStats *out) {
    if (!c || !out) return fail(ART_E_INVALID, "art_get_stats: null argument");
    if (c->traced && c->frame_ready) {
        int32_t r = use_device(c); if (r) return r;
        r = sync_all(c); if (r) return r;
        std::vector<uint32_t> raw(kCounterWords);
        if (c->fused && c->kind_primary == 8 && c->kind_shadow == 8) { // fused frames keep no counters: count from the frame's per-pixel bits + depth, here
            FrameSlot &S = c->slot[c->last];
            HIPC(hipMemsetAsync(S.d_counters.p, 0, kCounterWords * 4, c->stream_of(c->last)));
            if (c->n_local) { // of the frame the read calls refer to
                FrameArgs fa = make_frame_args(c, S, S.as_version);
                fa.pix_bits += (size_t)c->read_b * c->n_local; fa.depth += (size_t)c->read_b * c->W * c->H;
                launch_frame_stats(fa, S.d_counters.p, c->stream_of(c->last)); HIPC(hipGetLastError());
            }
            HIPC(hipStreamSynchronize(c->stream_of(c->last)));
        }
        HIPC(hipMemcpy(raw.data(), c->slot[c->last].d_counters.p, kCounterWords * 4, hipMemcpyDeviceToHost));
        uint64_t cnt[2] = {raw[0], raw[1]}; // folded totals (packet frames) + the slots (per-ray frames): one of the two is zero
        for (uint32_t k = 0; k < kSlotCount; k++) { cnt[0] += raw[kShadowSlots + k * kSlotStride]; cnt[1] += raw[kHitSlots + k * kSlotStride]; }
        uint64_t owned = 0; // pixels of owned tiles that fall inside the frame
        for (uint32_t t : c->tile_list) {
            uint32_t tx = t % c->tiles_x, ty = t / c->tiles_x;
            uint32_t w = (tx + 1) * kTile <= c->W ? kTile : c->W - tx * kTile, h = (ty + 1) * kTile <= c->H ? kTile : c->H - ty * kTile;
            owned += (uint64_t)w * h;
        }
        c->stats.primary_rays = owned; c->stats.shadow_rays = cnt[0]; c->stats.hit_pixels = cnt[1];
        c->stats.frame_launches = (c->fused && c->kind_primary == 8 && c->kind_shadow == 8) ? 1u : 4u;
        c->stats.split_blocks = c->plan.split1 + c->plan.split2;
        harvest_cost(c);
        c->stats.ao_rays = (uint64_t)c->ao_spp * cnt[1];
        if (c->ao_spp && c->slot[c->last].ao_valid) { float ams = 0; if (hipEventElapsedTime(&ams, c->slot[c->last].ao_ev[0], c->slot[c->last].ao_ev[1]) == hipSuccess) c->stats.ao_ms = ams; } // (only a slot whose latest frame had its AO pass has recorded these events: asking others leaves an error behind for the next hipGetLastError)
        float ms = 0;
        if (c->frame_no) {
            hipEvent_t *ev = c->ev[(c->frame_no - 1) % ArtContext::kRing];
            if (hipEventElapsedTime(&ms, ev[0], ev[4]) == hipSuccess) c->stats.frame_ms = ms;
            if (c->ev_fused[(c->frame_no - 1) % ArtContext::kRing]) { c->stats.trace_primary_ms = c->stats.frame_ms; c->stats.shade_ms = c->stats.trace_shadow_ms = c->stats.accumulate_ms = 0.f; }
            else {
            if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) c->stats.trace_primary_ms = ms;
            if (hipEventElapsedTime(&ms, ev[1], ev[2]) == hipSuccess) c->stats.shade_ms = ms;
            if (hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) c->stats.trace_shadow_ms = ms;
            if (hipEventElapsedTime(&ms, ev[3], ev[4]) == hipSuccess) c->stats.accumulate_ms = ms;
            }
        }
    }
    c->stats.first_move_ms = c->first_move_ms; c->stats.versions_ms = c->versions_ms;
    c->stats.refit_ms = c->last_refit_ms; c->stats.refit_cost_ratio = c->refit_cost_ratio; c->stats.refits = c->refits; c->stats.rebuilds = c->rebuilds;
    *out = c->stats;
    return ART_OK;
}

// ---- parity / debug surface ------------------------------------------------------------------------------------
int32_t art_read_hits(ArtContext *c, float *tuv, int32_t *ids, size_t n_pixels) {
    if (!c || !tuv || !ids) return fail(ART_E_INVALID, "art_read_hits: null argument");
    if (!c->traced) return fail(ART_E_STATE, "art_read_hits: nothing traced yet");
    if (n_pixels != (size_t)c->W * c->H) return fail(ART_E_INVALID, "art_read_hits: size mismatch");
    if (c->fused && c->kind_primary == 8 && c->kind_shadow == 8 && !(c->cfg.flags & ART_FLAG_KEEP_DEBUG)) return fail(ART_E_STATE, "art_read_hits: fused frames keep hit records only with ART_FLAG_KEEP_DEBUG");
    int32_t r = use_device(c); if (r) return r;
    r = sync_all(c); if (r) return r;
    std::vector<float4> h(c->n_local);
    std::vector<DevTri> tris(c->T);
    HIPC(hipMemcpy(h.data(), c->slot[c->last].d_hits.p + (size_t)c->read_b * c->n_local, (size_t)c->n_local * 16, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(tris.data(), c->bvh.tris, (size_t)c->T * sizeof(DevTri), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n_pixels; i++) { tuv[4 * i] = 0; tuv[4 * i + 1] = 0; tuv[4 * i + 2] = 0; tuv[4 * i + 3] = 0; ids[2 * i] = -2; ids[2 * i + 1] = -2; } // -2: not owned
    for (uint32_t p = 0; p < c->n_local; p++) {
        uint32_t tile = c->tile_list[p >> 10], q = p & 1023u, sub = q >> 6, l = q & 63u;
        uint32_t x = (tile % c->tiles_x) * kTile + (sub & 3u) * 8u + (l & 7u), y = (tile / c->tiles_x) * kTile + (sub >> 2) * 8u + (l >> 3);
        if (x >= c->W || y >= c->H) continue;
        size_t i = (size_t)y * c->W + x;
        uint32_t pos; std::memcpy(&pos, &h[p].w, 4);
        tuv[4 * i] = h[p].x; tuv[4 * i + 1] = h[p].y; tuv[4 * i + 2] = h[p].z;
        if (pos == kNoHit) { ids[2 * i] = -1; ids[2 * i + 1] = -1; }
        else { uint32_t gid; std::memcpy(&gid, &tris[pos].f[15], 4); gid_to_ids(c, gid, ids + 2 * i); }
    }
    return ART_OK;
}

int32_t art_read_shadow_bits(ArtContext *c, uint32_t *bits, size_t n_pixels) {
    if (!c || !bits) return fail(ART_E_INVALID, "art_read_shadow_bits: null argument");
    if (!(c->cfg.flags & ART_FLAG_KEEP_DEBUG)) return fail(ART_E_STATE, "art_read_shadow_bits: context created without ART_FLAG_KEEP_DEBUG");
    if (!c->traced) return fail(ART_E_STATE, "art_read_shadow_bits: nothing traced yet");
    if (n_pixels != (size_t)c->W * c->H) return fail(ART_E_INVALID, "art_read_shadow_bits: size mismatch");
    int32_t r = use_device(c); if (r) return r;
    r = sync_all(c); if (r) return r;
    std::vector<uint32_t> sb(c->n_local);
    HIPC(hipMemcpy(sb.data(), c->slot[c->last].d_shadow_bits.p + (size_t)c->read_b * c->n_local, (size_t)c->n_local * 4, hipMemcpyDeviceToHost));
    std::memset(bits, 0, n_pixels * 4);
    for (uint32_t p = 0; p < c->n_local; p++) {
        uint32_t tile = c->tile_list[p >> 10], q = p & 1023u, sub = q >> 6, l = q & 63u;
        uint32_t x = (tile % c->tiles_x) * kTile + (sub & 3u) * 8u + (l & 7u), y = (tile / c->tiles_x) * kTile + (sub >> 2) * 8u + (l >> 3);
        if (x >= c->W || y >= c->H) continue;
        bits[(size_t)y * c->W + x] = sb[p];
    }
    return ART_OK;
}

int32_t art_query_closest(ArtContext *c, const float *rays, uint32_t n, float *tuv, int32_t *ids) {
    if (!c || (n && (!rays || !tuv || !ids))) return fail(ART_E_INVALID, "art_query_closest: null argument");
    if (!c->built) return fail(ART_E_STATE, "art_query_closest: scene not built");
    if (n == 0) return ART_OK;
    int32_t r = use_device(c); if (r) return r;
    float4 *d_r = nullptr, *d_h = nullptr;
    HIPC(hipMalloc(&d_r, (size_t)n * 32));
    hipError_t e = hipMalloc(&d_h, (size_t)n * 16);
    if (e != hipSuccess) { (void)hipFree(d_r); return hipfail(e, "hipMalloc"); }
    std::vector<float4> h(n);
    std::vector<DevTri> tris(c->T);
    e = hipMemcpy(d_r, rays, (size_t)n * 32, hipMemcpyHostToDevice);
    const int qkind = c->kind_primary == 8 ? 2 : c->kind_primary;
    if (e == hipSuccess && (refresh_now(c) != ART_OK || ensure_wide(c, true) != ART_OK || ensure_binary(c, qkind == 2) != ART_OK)) e = hipErrorUnknown; // (a pending move is applied first)
    if (e == hipSuccess) e = c->slot[0].d_counters.ensure(kCounterWords);
    if (e == hipSuccess) e = hipMemsetAsync(c->slot[0].d_counters.p, 0, kCounterWords * 4, c->main_stream());
    if (e == hipSuccess) { const AsPtrs as = as_ptrs(c, c->as_cur); launch_query_closest(BvhView{c->bvh.nodes, as.wide, as.tris, qkind, TraceTune{c->tuning.trace_chunk, c->tuning.trace_refill, c->tuning.trace_blocks, c->tuning.trace_leaf_batch}}, d_r, n, d_h, c->slot[0].d_counters.p + 64 + 512, c->main_stream()); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipStreamSynchronize(c->main_stream());
    if (e == hipSuccess) e = hipMemcpy(h.data(), d_h, (size_t)n * 16, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(tris.data(), c->bvh.tris, (size_t)c->T * sizeof(DevTri), hipMemcpyDeviceToHost);
    (void)hipFree(d_r); (void)hipFree(d_h);
    if (e != hipSuccess) return hipfail(e, "art_query_closest");
    for (uint32_t i = 0; i < n; i++) {
        uint32_t pos; std::memcpy(&pos, &h[i].w, 4);
        tuv[4 * i] = h[i].x; tuv[4 * i + 1] = h[i].y; tuv[4 * i + 2] = h[i].z; tuv[4 * i + 3] = 0;
        if (pos == kNoHit) { ids[2 * i] = -1; ids[2 * i + 1] = -1; }
        else { uint32_t gid; std::memcpy(&gid, &tris[pos].f[15], 4); gid_to_ids(c, gid, ids + 2 * i); }
    }
    return ART_OK;
}

int32_t art_query_any(ArtContext *c, const float *rays, uint32_t n, uint8_t *hit) {
    if (!c || (n && (!rays || !hit))) return fail(ART_E_INVALID, "art_query_any: null argument");
    if (!c->built) return fail(ART_E_STATE, "art_query_any: scene not built");
    if (n == 0) return ART_OK;
    int32_t r = use_device(c); if (r) return r;
    float4 *d_r = nullptr; uint32_t *d_h = nullptr;
    HIPC(hipMalloc(&d_r, (size_t)n * 32));
    hipError_t e = hipMalloc(&d_h, (size_t)n * 4);
    if (e != hipSuccess) { (void)hipFree(d_r); return hipfail(e, "hipMalloc"); }
    std::vector<uint32_t> h(n);
    e = hipMemcpy(d_r, rays, (size_t)n * 32, hipMemcpyHostToDevice);
    const int qkind = c->kind_shadow == 8 ? 4 : c->kind_shadow;
    if (e == hipSuccess && (refresh_now(c) != ART_OK || ensure_wide(c, true) != ART_OK || ensure_binary(c, qkind == 2) != ART_OK)) e = hipErrorUnknown;
    if (e == hipSuccess) e = c->slot[0].d_counters.ensure(kCounterWords);
    if (e == hipSuccess) e = hipMemsetAsync(c->slot[0].d_counters.p, 0, kCounterWords * 4, c->main_stream());
    if (e == hipSuccess) { const AsPtrs as = as_ptrs(c, c->as_cur); launch_query_any(BvhView{c->bvh.nodes, as.wide, as.tris, qkind, TraceTune{c->tuning.trace_chunk, c->tuning.trace_refill, c->tuning.trace_blocks, c->tuning.trace_leaf_batch}}, d_r, n, d_h, c->slot[0].d_counters.p + 64 + 512, c->main_stream()); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipStreamSynchronize(c->main_stream());
    if (e == hipSuccess) e = hipMemcpy(h.data(), d_h, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_r); (void)hipFree(d_h);
    if (e != hipSuccess) return hipfail(e, "art_query_any");
    for (uint32_t i = 0; i < n; i++) hit[i] = (uint8_t)h[i];
    return ART_OK;
}

int32_t art_get_lbvh(ArtContext *c, uint32_t *leaf_gid, uint64_t *keys, int32_t *child, float *node_lo, float *node_hi, float *leaf_lo, float *leaf_hi) {
    if (!c) return fail(ART_E_INVALID, "art_get_lbvh: null context");
    if (!c->built) return fail(ART_E_STATE, "art_get_lbvh: scene not built");
    int32_t r = use_device(c); if (r) return r;
    r = refresh_now(c); if (r) return r;       // after a move: the boxes of where the models are now (the keys and the topology are the build's)
    if (!c->bvh.canon_boxes) c->binary_epoch = ~0ull;   // the build left the canonical tree's boxes for now: have them made
    r = ensure_binary(c, true); if (r) return r;
    size_t T = c->T, NI = T > 1 ? T - 1 : 0;
    if (leaf_gid) HIPC(hipMemcpy(leaf_gid, c->bvh.leaf_gid, T * 4, hipMemcpyDeviceToHost));
    if (keys) HIPC(hipMemcpy(keys, c->bvh.keys, T * 8, hipMemcpyDeviceToHost));
    if (child && NI) HIPC(hipMemcpy(child, c->bvh.child, NI * 8, hipMemcpyDeviceToHost));
    if (node_lo && NI) HIPC(hipMemcpy(node_lo, c->bvh.node_lo, NI * 12, hipMemcpyDeviceToHost));
    if (node_hi && NI) HIPC(hipMemcpy(node_hi, c->bvh.node_hi, NI * 12, hipMemcpyDeviceToHost));
    if (leaf_lo) HIPC(hipMemcpy(leaf_lo, c->bvh.leaf_lo, T * 12, hipMemcpyDeviceToHost));
    if (leaf_hi) HIPC(hipMemcpy(leaf_hi, c->bvh.leaf_hi, T * 12, hipMemcpyDeviceToHost));
    return ART_OK;
}

// the tree the walks actually use: the SAH topology when it was built (default), else the canonical one; leaves are those of art_get_lbvh
int32_t art_get_traversal_tree(ArtContext *c, int32_t *child, float *node_lo, float *node_hi) {
    if (!c) return fail(ART_E_INVALID, "art_get_traversal_tree: null context");
    if (!c->built) return fail(ART_E_STATE, "art_get_traversal_tree: scene not built");
    int32_t r = use_device(c); if (r) return r;
    r = refresh_now(c); if (r) return r;
    r = ensure_binary(c, true); if (r) return r;
    size_t T = c->T, NI = T > 1 ? T - 1 : 0;
    const bool sah = c->bvh.trav_child != nullptr;
    if (child && NI) HIPC(hipMemcpy(child, sah ? c->bvh.trav_child : c->bvh.child, NI * 8, hipMemcpyDeviceToHost));
    if (node_lo && NI) HIPC(hipMemcpy(node_lo, sah ? c->bvh.trav_lo : c->bvh.node_lo, NI * 12, hipMemcpyDeviceToHost));
    if (node_hi && NI) HIPC(hipMemcpy(node_hi, sah ? c->bvh.trav_hi : c->bvh.node_hi, NI * 12, hipMemcpyDeviceToHost));
    return ART_OK;
}

// the 4-wide collapse of that tree, as the walks read it: n_nodes records of 64 B (quantised, per-ray walks) and of 128 B (float boxes, packet walks).
// Builds it if no walk has needed it yet.  Either pointer may be NULL; *n_nodes is always set.
int32_t art_get_wide_nodes(ArtContext *c, void *quantised, void *floats, size_t capacity_nodes, uint32_t *n_nodes) {
    if (!c || !n_nodes) return fail(ART_E_INVALID, "art_get_wide_nodes: null argument");
    if (!c->built) return fail(ART_E_STATE, "art_get_wide_nodes: scene not built");
    int32_t r = use_device(c); if (r) return r;
    r = refresh_now(c); if (r) return r;
    r = ensure_wide(c, true); if (r) return r;
    *n_nodes = c->bvh.n_wide;
    if ((quantised || floats) && capacity_nodes < c->bvh.n_wide) return fail(ART_E_INVALID, "art_get_wide_nodes: buffers too small");
    const AsPtrs as = as_ptrs(c, c->as_cur);   // the version the next frame would read
    if (quantised) HIPC(hipMemcpy(quantised, as.wide, (size_t)c->bvh.n_wide * sizeof(DevNode4), hipMemcpyDeviceToHost));
    if (floats) HIPC(hipMemcpy(floats, as.widef, (size_t)c->bvh.n_wide * sizeof(DevNodeW), hipMemcpyDeviceToHost));
    return ART_OK;
}

} // extern "C"
