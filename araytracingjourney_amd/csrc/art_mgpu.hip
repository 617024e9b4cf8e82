// art_mgpu.hip -- the sharded frame as one C-ABI surface (include/art.h, art_mgpu_*): trace this rank's share, gather the compact colour
// tiles of a group of launches (RCCL over xGMI), un-tile the group with one launch.  Two placements of the assembled frames:
//   ART_MGPU_ROOT_RANK0   every frame is assembled on rank 0: ONE ncclGather per group;
//   ART_MGPU_ROOT_SPREAD  frame f is assembled on rank f mod world: per group ONE ncclGroupStart .. ncclGroupEnd of ncclSend / ncclRecv -- each frame's
//                         gather, all of them at once.  xGMI is point to point (one link per pair of GPUs): a single root receives through its 7 links
//                         while the 42 others idle, and with RGB32F tiles that, not the tracing, bounds the job (2 GPUs: 12.4 MB per frame over ONE
//                         link = 200 us against 150 us for the whole frame on one GPU); spread roots load every link in both directions alike.
// New functionality of BASELINE.json's north_star; the reference renders on one queue of one device (renderer.rs:188) and has no counterpart.
//
// Everything here is host orchestration over the context's public entry points (art_trace, art_frames_done, art_bind_color_tiles_ring,
// art_untile_gathered_frames); the rules it follows were measured in round 1 (profiles/README.md r1i, r1n):
//   - the exchange stream carries no device-side wait: a group is SUBMITTED once the host has seen its frames finish (art_frames_done).
//     A hipStreamWaitEvent in front of each collective took ~40 us to retire on a GPU whose other queues are busy;
//   - a frame never waits on the device for the exchange that read its slot's previous tiles either (a cross-stream wait in front of a
//     launch makes it acquire at system scope and costs the frames in flight their L2 contents): every slot has several tile buffers,
//     written in turn, and the host checks (almost always: finds) the old exchange finished before it launches;
//   - the slots of a group are contiguous in memory, so a group is one message per peer.
// RCCL is resolved with dlopen at art_mgpu_create: libart.so itself does not depend on it.
#include "art_internal.h"
#include <dlfcn.h>
#include <deque>
#include <new>

using namespace art;

namespace {

// errors of this file reach the caller through art_last_error() like every other entry point's: art_api.hip owns that string (set_last_error)
int32_t mg_fail(int32_t code, const std::string &msg) { set_last_error(msg.c_str()); return code; }

// ---- the RCCL entry points this file uses (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGetErrorString, ncclGather, ncclSend, ncclRecv,
// ncclGroupStart, ncclGroupEnd), resolved at run time ----------------------------------------------------------------------------------------
struct RcclId { char internal[ART_MGPU_ID_BYTES]; };
static_assert(sizeof(RcclId) == 128, "ncclUniqueId is 128 bytes (rccl.h:40)");
struct Rccl {
    void *so = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(void **comm, int nranks, RcclId id, int rank) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*Gather)(const void *send, void *recv, size_t count, int datatype, int root, void *comm, hipStream_t stream) = nullptr;
    int (*Send)(const void *send, size_t count, int datatype, int peer, void *comm, hipStream_t stream) = nullptr;
    int (*Recv)(void *recv, size_t count, int datatype, int peer, void *comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool ok() const { return GetUniqueId && CommInitRank && CommDestroy && GetErrorString && Gather && Send && Recv && GroupStart && GroupEnd; }
};
constexpr int kNcclUint8 = 1; // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)
Rccl &rccl() {
    static Rccl r;
    if (!r.so) {
        for (const char *name : {"librccl.so.1", "librccl.so"}) { r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (r.so) break; }
        if (r.so) {
            r.GetUniqueId = (int (*)(RcclId *))dlsym(r.so, "ncclGetUniqueId");
            r.CommInitRank = (int (*)(void **, int, RcclId, int))dlsym(r.so, "ncclCommInitRank");
            r.CommDestroy = (int (*)(void *))dlsym(r.so, "ncclCommDestroy");
            r.GetErrorString = (const char *(*)(int))dlsym(r.so, "ncclGetErrorString");
            r.Gather = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(r.so, "ncclGather");
            r.Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))dlsym(r.so, "ncclSend");
            r.Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))dlsym(r.so, "ncclRecv");
            r.GroupStart = (int (*)())dlsym(r.so, "ncclGroupStart");
            r.GroupEnd = (int (*)())dlsym(r.so, "ncclGroupEnd");
        }
    }
    return r;
}

enum SlotState : uint8_t { kFree = 0, kQueued = 1, kSent = 2 }; // tile buffer of a slot: never exchanged / in a group not yet submitted / exchange submitted (event recorded)
struct Group { uint32_t k0, n, buf; uint64_t first_launch, first_frame; };   // ring slots [k0, k0 + n) of tile buffer `buf`; the context's launch numbers first_launch ..; the job's frame number of its first frame

} // namespace

struct ArtMgpu {
    ArtContext *ctx = nullptr;
    ArtMgpuConfig cfg{};
    ArtLayout lay{};
    uint32_t F = 1, B = 1, GB = 1, NBUF = 4, G = 1;   // ring slots, frames per launch, launches per gather, tile buffers per slot, shards of the frame
    bool renders = true, root = false;                 // root: this rank assembles frames (rank 0; every rank with spread roots)
    bool spread = false; uint32_t nf_cap = 0;          // ART_MGPU_ROOT_SPREAD; frames of one group a rank can be the root of
    char *stage = nullptr;                             // spread roots over the host exchange: one frame's tiles of every rank, as the hook leaves them
    size_t slot_bytes = 0, frame_bytes = 0;            // compact tiles of one launch (B frames); one assembled frame
    char *tiles = nullptr;                             // [NBUF][F][slot_bytes]
    char *gathered = nullptr;                          // a root: [world][GB][slot_bytes] (spread: [world][nf_cap][one frame's tiles]), one group at a time
    char *frames = nullptr;                            // a root: [GB * B][frame_bytes] (spread: [nf_cap]), the group un-tiled last
    hipStream_t xs = nullptr;                          // the exchange: gathers + un-tiles, in submission order
    SlotState state[kTileRingMax][kMaxFrameSlots] = {};
    uint32_t ev_of[kTileRingMax][kMaxFrameSlots] = {};
    std::vector<hipEvent_t> events; uint32_t next_event = 0; // one per exchange, shared by the group's slots; reused long after every slot of it was re-assigned
    std::deque<Group> fifo;                            // groups whose frames may still be running
    uint64_t traced = 0, gathers = 0;                  // launches traced through this object; exchanges submitted
    uint32_t pend_k0 = 0, pend_n = 0;
    uint32_t newest = 0; bool have_frame = false;      // where in `frames` the most recent frame sits
    void *comm = nullptr;
    int32_t failed = ART_OK; std::string failed_msg;   // an exchange failed: its group's frames are lost and the ranks are out of step -- every later call reports it
};

namespace {

#define MGH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return mg_fail(ART_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define MGA(x) do { int32_t r_ = (x); if (r_ != ART_OK) return r_; } while (0) /* a libart call: its message is already in art_last_error */

// spread roots: the group's frames lie back to back in the send buffer (a launch's B frames, launch after launch); frame j goes to rank
// (first_frame + j) mod world, which receives it from every rank (itself included) into [rank][its i-th frame of the group]
int32_t run_exchange_spread(ArtMgpu *m, const Group &g, const char *send) {
    const uint32_t W = m->cfg.world, r = m->cfg.rank, nf = g.n * m->B;
    const size_t ftb = m->slot_bytes / m->B;           // one frame's compact tiles of one rank
    const bool nccl = m->cfg.transport == ART_MGPU_RCCL;
    uint32_t mine = 0;
    int rc = 0;
    bool overflow = false;
    if (nccl && (rc = rccl().GroupStart()) != 0) return mg_fail(ART_E_HIP, std::string("ncclGroupStart: ") + rccl().GetErrorString(rc));
    for (uint32_t j = 0; j < nf; j++) {
        const uint32_t q = (uint32_t)((g.first_frame + j) % W);
        if (q == r && mine >= m->nf_cap) { overflow = true; break; }   // (cannot happen: nf_cap = ceil(GB * B / world); checked so that no receive can land outside the buffer)
        if (nccl) {
            if ((rc = rccl().Send(send + (size_t)j * ftb, ftb, kNcclUint8, (int)q, m->comm, m->xs)) != 0) break;
            if (q == r) for (uint32_t p = 0; p < W && rc == 0; p++) rc = rccl().Recv(m->gathered + ((size_t)p * m->nf_cap + mine) * ftb, ftb, kNcclUint8, (int)p, m->comm, m->xs);
            if (rc != 0) break;
        } else {
            int32_t hr = m->cfg.exchange(m->cfg.exchange_user, send + (size_t)j * ftb, ftb, q == r ? m->stage : nullptr, q, m->xs);
            if (hr != 0) return mg_fail(ART_E_HIP, "art_mgpu: the host exchange function failed (" + std::to_string(hr) + ")");
            if (q == r) { // the hook leaves [rank][ftb]; the un-tile wants a rank's frames of the group back to back
                MGH(hipMemcpy2DAsync(m->gathered + (size_t)mine * ftb, (size_t)m->nf_cap * ftb, m->stage, ftb, ftb, W, hipMemcpyDeviceToDevice, m->xs));
                MGH(hipStreamSynchronize(m->xs)); // the next call of the hook overwrites the staging area, and a hook need not be ordered on xs while it runs
            }
        }
        if (q == r) mine++;
    }
    if (nccl) {
        int rc2 = rccl().GroupEnd();   // (always closed, also after a failed call inside it)
        if (rc != 0 || rc2 != 0) return mg_fail(ART_E_HIP, std::string("ncclSend / ncclRecv / ncclGroupEnd: ") + rccl().GetErrorString(rc ? rc : rc2));
    }
    if (overflow) return mg_fail(ART_E_STATE, "art_mgpu: more frames of a group fall to this rank than it has room for");
    if (mine) {
        MGA(art_untile_gathered_frames(m->ctx, m->gathered, W, m->nf_cap * m->lay.tiles_padded, mine, m->frames, m->xs));
        m->newest = mine - 1; m->have_frame = true;
    }
    return ART_OK;
}

int32_t run_exchange(ArtMgpu *m, const Group &g) {
    const char *send = m->tiles + ((size_t)g.buf * m->F + g.k0) * m->slot_bytes;
    const size_t bytes = (size_t)g.n * m->slot_bytes;
    if (m->spread) MGA(run_exchange_spread(m, g, send));
    else if (m->cfg.transport == ART_MGPU_RCCL) {
        int rc = rccl().Gather(send, m->root ? m->gathered : nullptr, bytes, kNcclUint8, 0, m->comm, m->xs);
        if (rc != 0) return mg_fail(ART_E_HIP, std::string("ncclGather: ") + rccl().GetErrorString(rc));
    } else {
        if (m->root) MGH(hipStreamSynchronize(m->xs)); // the un-tile of the previous group may still read `gathered`, and a hook need not order its writes behind xs (a blocking copy on the null stream does not)
        int32_t rc = m->cfg.exchange(m->cfg.exchange_user, send, bytes, m->root ? m->gathered : nullptr, 0u, m->xs);
        if (rc != 0) return mg_fail(ART_E_HIP, "art_mgpu: the host exchange function failed (" + std::to_string(rc) + ")");
    }
    if (m->root && !m->spread) { // rank r's block holds the group's launches back to back: a frame's shards are n * B * padded tiles apart; with a dedicated compositor shard s came from rank s + 1
        const uint32_t padded = m->lay.tiles_padded;
        const char *first = m->gathered + (m->cfg.compositor == ART_MGPU_DEDICATED ? bytes : 0);
        MGA(art_untile_gathered_frames(m->ctx, first, m->G, g.n * m->B * padded, g.n * m->B, m->frames, m->xs));
        m->newest = g.n * m->B - 1; m->have_frame = true;
    }
    const uint32_t e = m->next_event; m->next_event = (m->next_event + 1) % (uint32_t)m->events.size();
    MGH(hipEventRecord(m->events[e], m->xs));
    for (uint32_t k = g.k0; k < g.k0 + g.n; k++) { m->state[g.buf][k] = kSent; m->ev_of[g.buf][k] = e; }
    m->gathers++;
    return ART_OK;
}

int32_t sticky(ArtMgpu *m) { return m->failed ? mg_fail(m->failed, "art_mgpu: an earlier exchange failed (" + m->failed_msg + "); destroy this object") : ART_OK; }

// submit, in order, every queued group whose frames the host can see finished; force: wait for them
int32_t poll(ArtMgpu *m, bool force) {
    MGA(sticky(m));
    while (!m->fifo.empty()) {
        const Group g = m->fifo.front();
        if (m->renders) {
            int32_t done = 0;
            MGA(art_frames_done(m->ctx, g.first_launch, g.n, &done, nullptr));
            if (!done) {
                if (!force) return ART_OK;
                // wait for THIS group's frames only (the host spins on their events): at a flush behind a short burst of launches the first group's
                // exchange then runs while the later launches still trace, instead of every exchange queueing up behind the last frame (art_sync)
                while (!done) MGA(art_frames_done(m->ctx, g.first_launch, g.n, &done, nullptr));
            }
        }
        m->fifo.pop_front();
        const int32_t r = run_exchange(m, g);
        if (r != ART_OK) { // the group is gone and its slots will never be sent: free them, and latch the error -- a later trace must not wait for them, a flush not report success
            for (uint32_t k = g.k0; k < g.k0 + g.n; k++) m->state[g.buf][k] = kFree;
            m->failed = r; m->failed_msg = art_last_error();
            return r;
        }
    }
    return ART_OK;
}

int32_t close_group(ArtMgpu *m, bool force) {
    if (m->pend_n) {
        const uint32_t buf = (uint32_t)(((m->traced - 1) / m->F) % m->NBUF); // the tile buffer these launches wrote
        uint64_t launches = 0;
        if (m->renders) { int32_t d; MGA(art_frames_done(m->ctx, 0, 0, &d, &launches)); }
        m->fifo.push_back(Group{m->pend_k0, m->pend_n, buf, m->renders ? launches - m->pend_n : 0, (m->traced - m->pend_n) * m->B});
        for (uint32_t k = m->pend_k0; k < m->pend_k0 + m->pend_n; k++) m->state[buf][k] = kQueued;
        m->pend_k0 = (m->pend_k0 + m->pend_n) % m->F; m->pend_n = 0;
    }
    return poll(m, force);
}

} // namespace

extern "C" {

int32_t art_mgpu_shard(uint32_t rank, uint32_t world, uint32_t compositor, uint32_t *shard_rank, uint32_t *shard_count) {
    if (!shard_rank || !shard_count || world == 0 || rank >= world || compositor > 1) return mg_fail(ART_E_INVALID, "art_mgpu_shard: bad argument");
    if (compositor == ART_MGPU_DEDICATED) {
        if (world < 2) return mg_fail(ART_E_INVALID, "art_mgpu_shard: a dedicated compositor needs at least two ranks");
        *shard_count = world - 1; *shard_rank = rank ? rank - 1 : 0;   // rank 0 keeps a context for the layout tables and the un-tile; it traces nothing
    } else { *shard_count = world; *shard_rank = rank; }
    return ART_OK;
}

int32_t art_mgpu_unique_id(uint8_t id[ART_MGPU_ID_BYTES]) {
    if (!id) return mg_fail(ART_E_INVALID, "art_mgpu_unique_id: null argument");
    if (!rccl().ok()) return mg_fail(ART_E_NO_DEVICE, "art_mgpu_unique_id: librccl.so.1 is not loadable (RCCL is the only built-in transport)");
    RcclId u;
    int rc = rccl().GetUniqueId(&u);
    if (rc != 0) return mg_fail(ART_E_HIP, std::string("ncclGetUniqueId: ") + rccl().GetErrorString(rc));
    std::memcpy(id, u.internal, ART_MGPU_ID_BYTES);
    return ART_OK;
}

int32_t art_mgpu_destroy(ArtMgpu *m) {
    if (!m) return ART_OK;
    if (m->ctx) { (void)art_sync(m->ctx); for (uint32_t k = 0; k < m->F; k++) (void)art_bind_color_tiles(m->ctx, k, nullptr, 0); }
    if (m->xs) (void)hipStreamSynchronize(m->xs);
    if (m->comm) (void)rccl().CommDestroy(m->comm);
    for (hipEvent_t e : m->events) (void)hipEventDestroy(e);
    if (m->xs) (void)hipStreamDestroy(m->xs);
    (void)hipFree(m->tiles); (void)hipFree(m->gathered); (void)hipFree(m->frames); (void)hipFree(m->stage);
    delete m;
    return ART_OK;
}

int32_t art_mgpu_create(ArtContext *ctx, const ArtMgpuConfig *cfg, const uint8_t id[ART_MGPU_ID_BYTES], ArtMgpu **out) {
    if (!ctx || !cfg || !out) return mg_fail(ART_E_INVALID, "art_mgpu_create: null argument");
    *out = nullptr;
    if (cfg->world == 0 || cfg->rank >= cfg->world || cfg->compositor > 1 || cfg->transport > 1 || cfg->tile_buffers > kTileRingMax || cfg->roots > 1)
        return mg_fail(ART_E_INVALID, "art_mgpu_create: bad rank / world / compositor / transport / tile_buffers / roots");
    if (cfg->roots == ART_MGPU_ROOT_SPREAD && cfg->compositor == ART_MGPU_DEDICATED) return mg_fail(ART_E_INVALID, "art_mgpu_create: spread roots and a dedicated compositor exclude each other (every rank traces and assembles)");
    if (cfg->transport == ART_MGPU_HOST_EXCHANGE && !cfg->exchange) return mg_fail(ART_E_INVALID, "art_mgpu_create: ART_MGPU_HOST_EXCHANGE without an exchange function");
    if (cfg->transport == ART_MGPU_RCCL && !id) return mg_fail(ART_E_INVALID, "art_mgpu_create: the RCCL transport needs the job's id (art_mgpu_unique_id on rank 0)");
    uint32_t sr = 0, sc = 0;
    MGA(art_mgpu_shard(cfg->rank, cfg->world, cfg->compositor, &sr, &sc));
    ArtMgpu *m = new (std::nothrow) ArtMgpu();
    if (!m) return mg_fail(ART_E_NOMEM, "art_mgpu_create: out of memory");
    m->ctx = ctx; m->cfg = *cfg;
    int32_t r = art_get_layout(ctx, &m->lay);
    if (r) { m->ctx = nullptr; art_mgpu_destroy(m); return r; }
    auto bail = [&](int32_t code) { m->ctx = nullptr; art_mgpu_destroy(m); return code; };
    if (m->lay.shard_count != sc || m->lay.shard_rank != sr)
        return bail(mg_fail(ART_E_INVALID, "art_mgpu_create: the context's shard is not the one art_mgpu_shard gives this rank (" + std::to_string(sr) + " of " + std::to_string(sc) + ")"));
    if (m->lay.tile_bytes == 0 || m->lay.tiles_padded == 0) return bail(mg_fail(ART_E_STATE, "art_mgpu_create: the context writes no compact tiles (one shard: create it with ART_FLAG_TILE_OUTPUT)"));
    m->F = m->lay.frames_in_flight; m->B = m->lay.frames_per_launch; m->G = sc;
    m->NBUF = cfg->tile_buffers ? cfg->tile_buffers : 4;
    m->GB = cfg->launches_per_gather && cfg->launches_per_gather < m->F ? cfg->launches_per_gather : m->F;
    while (m->F % m->GB) m->GB--;                      // whole groups per trip round the ring: a group is one contiguous slice
    m->spread = cfg->roots == ART_MGPU_ROOT_SPREAD;
    m->root = cfg->rank == 0 || m->spread;
    m->nf_cap = (m->GB * m->B + cfg->world - 1) / cfg->world;
    m->renders = !(cfg->compositor == ART_MGPU_DEDICATED && cfg->rank == 0);
    m->slot_bytes = (size_t)m->B * m->lay.tiles_padded * m->lay.tile_bytes;
    m->frame_bytes = (size_t)m->lay.width * m->lay.height * (m->lay.tile_bytes == 4u * kTilePixels ? 4u : 16u);   // the assembled frame: B10G11R11 words, or RGBA32F (the tiles carry RGB; alpha is the constant 1)
    hipError_t e = hipMalloc(&m->tiles, (size_t)m->NBUF * m->F * m->slot_bytes);
    if (e == hipSuccess) e = hipMemset(m->tiles, 0, (size_t)m->NBUF * m->F * m->slot_bytes);
    const size_t n_frames_kept = m->spread ? m->nf_cap : (size_t)m->GB * m->B;
    if (e == hipSuccess && m->root) e = hipMalloc(&m->gathered, m->spread ? (size_t)cfg->world * m->nf_cap * (m->slot_bytes / m->B) : (size_t)cfg->world * m->GB * m->slot_bytes);
    if (e == hipSuccess && m->root) e = hipMalloc(&m->frames, n_frames_kept * m->frame_bytes);
    if (e == hipSuccess && m->root) e = hipMemset(m->frames, 0, n_frames_kept * m->frame_bytes);
    if (e == hipSuccess && m->spread && cfg->transport == ART_MGPU_HOST_EXCHANGE) e = hipMalloc(&m->stage, (size_t)cfg->world * (m->slot_bytes / m->B));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->xs, hipStreamNonBlocking);
    m->events.assign((size_t)m->NBUF * m->F + 8, nullptr);
    for (size_t i = 0; i < m->events.size() && e == hipSuccess; i++) e = hipEventCreateWithFlags(&m->events[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return bail(mg_fail(ART_E_HIP, std::string("art_mgpu_create: ") + hipGetErrorString(e)));
    if (cfg->transport == ART_MGPU_RCCL) {
        if (!rccl().ok()) return bail(mg_fail(ART_E_NO_DEVICE, "art_mgpu_create: librccl.so.1 is not loadable"));
        RcclId u; std::memcpy(u.internal, id, ART_MGPU_ID_BYTES);
        int rc = rccl().CommInitRank(&m->comm, (int)cfg->world, u, (int)cfg->rank);   // on the context's device: art_get_layout made it current
        if (rc != 0) { m->comm = nullptr; return bail(mg_fail(ART_E_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(rc))); }
    }
    for (uint32_t k = 0; k < m->F; k++) {              // slot k renders into tiles[trip % NBUF][k]
        void *bufs[kTileRingMax];
        for (uint32_t b = 0; b < m->NBUF; b++) bufs[b] = m->tiles + ((size_t)b * m->F + k) * m->slot_bytes;
        r = art_bind_color_tiles_ring(ctx, k, bufs, m->NBUF, m->slot_bytes);
        if (r) { for (uint32_t j = 0; j <= k; j++) (void)art_bind_color_tiles(ctx, j, nullptr, 0); return bail(r); }   // no slot keeps a pointer into the buffers bail() frees
    }
    // Ring slot, tile buffer and group boundaries follow the context's launch count, and every rank must cut its groups alike (a group is one
    // collective): whatever the caller traced before, the count starts again from zero here, on every rank.
    r = ring_rewind(ctx);
    if (r) { for (uint32_t k = 0; k < m->F; k++) (void)art_bind_color_tiles(ctx, k, nullptr, 0); return bail(r); }   // (bail() frees the buffers the slots were just bound to)
    *out = m;
    return ART_OK;
}

int32_t art_mgpu_trace(ArtMgpu *m) {
    if (!m) return mg_fail(ART_E_INVALID, "art_mgpu_trace: null argument");
    MGA(sticky(m));
    const uint32_t k = (uint32_t)(m->traced % m->F), buf = (uint32_t)((m->traced / m->F) % m->NBUF);   // the ring slot and the tile buffer this launch takes (art_trace: tiles_for)
    while (m->state[buf][k] == kQueued) {                         // its previous contents have not even been sent: NBUF trips behind, rare
        if (m->fifo.empty()) return mg_fail(ART_E_STATE, "art_mgpu_trace: a tile buffer is queued for an exchange no group holds");   // (cannot happen; never spin on it)
        MGA(poll(m, false));
    }
    if (m->state[buf][k] == kSent) {
        // Gate on the HOST: the event is NBUF trips old and has almost always fired.  (A cross-stream wait queued in front of the launch would
        // cost the frames in flight their L2 contents: 115 instead of 55 us per frame on a 1/8 share, profiles/README.md r1n.)
        hipEvent_t ev = m->events[m->ev_of[buf][k]];
        if (hipEventQuery(ev) != hipSuccess) MGH(hipEventSynchronize(ev));
        m->state[buf][k] = kFree;
    }
    if (m->renders) {
        uint32_t next = 0;
        MGA(art_frames_in_flight(m->ctx, nullptr, &next));
        if (next != k) return mg_fail(ART_E_STATE, "art_mgpu_trace: the context was traced behind this object's back (ring slots out of step)");
        MGA(art_trace(m->ctx));
    }
    m->traced++; m->pend_n++;
    if (m->pend_n == m->GB || k + 1 == m->F) return close_group(m, false);   // one exchange per GB launches, never across the ring's wrap
    if (!m->fifo.empty() && m->traced % 4 == 0) return poll(m, false);
    return ART_OK;
}

int32_t art_mgpu_flush(ArtMgpu *m) {
    if (!m) return mg_fail(ART_E_INVALID, "art_mgpu_flush: null argument");
    MGA(sticky(m));
    MGA(close_group(m, true));
    MGH(hipStreamSynchronize(m->xs));
    return ART_OK;
}

int32_t art_mgpu_device_frame(ArtMgpu *m, void **dev_ptr, size_t *bytes) {
    if (!m || !dev_ptr) return mg_fail(ART_E_INVALID, "art_mgpu_device_frame: null argument");
    if (!m->root) return mg_fail(ART_E_STATE, "art_mgpu_device_frame: only rank 0 holds the assembled frame");
    if (!m->have_frame) return mg_fail(ART_E_STATE, "art_mgpu_device_frame: nothing gathered yet (with spread roots: no frame has fallen to this rank)");
    *dev_ptr = m->frames + (size_t)m->newest * m->frame_bytes;
    if (bytes) *bytes = m->frame_bytes;
    return ART_OK;
}

int32_t art_mgpu_read_frame(ArtMgpu *m, void *dst, size_t bytes) {
    void *p = nullptr; size_t n = 0;
    if (!dst) return mg_fail(ART_E_INVALID, "art_mgpu_read_frame: null argument");
    MGA(art_mgpu_device_frame(m, &p, &n));
    if (bytes != n) return mg_fail(ART_E_INVALID, "art_mgpu_read_frame: size mismatch");
    MGH(hipStreamSynchronize(m->xs));
    MGH(hipMemcpy(dst, p, n, hipMemcpyDeviceToHost));
    return ART_OK;
}

int32_t art_mgpu_counts(ArtMgpu *m, uint64_t *launches_traced, uint64_t *gathers, uint32_t *launches_per_gather) {
    if (!m) return mg_fail(ART_E_INVALID, "art_mgpu_counts: null argument");
    if (launches_traced) *launches_traced = m->traced;
    if (gathers) *gathers = m->gathers;
    if (launches_per_gather) *launches_per_gather = m->GB;
    return ART_OK;
}

int32_t art_mgpu_pending(ArtMgpu *m, uint32_t *groups_queued, uint32_t *launches_open) {
    if (!m) return mg_fail(ART_E_INVALID, "art_mgpu_pending: null argument");
    if (groups_queued) *groups_queued = (uint32_t)m->fifo.size();
    if (launches_open) *launches_open = m->pend_n;
    return ART_OK;
}

} // extern "C"
