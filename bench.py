#!/usr/bin/env python3
"""bench.py -- Mray/s (primary + shadow) on BASELINE.json configs[1]: Sponza-class scene, 1920x1080, one directional
light, one shadow ray per lit pixel.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one frame of the hot path (primary rays + closest hit, hit reconstruction + PBR direct light, shadow rays,
accumulation; for N > 1 also the RCCL gather of the HDR tiles to rank 0 and the un-tile).  The scene, BVH, camera and
lights are resident in HBM before the timed region.  N > 1 shards the frame by 32x32 screen tile (strong scaling) through
libart's art_mgpu_* entry points: the whole sharded frame -- trace, ncclGather, un-tile -- is C ABI, this file is its caller.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
# Several frames are kept in flight (the reference keeps 3: renderer.rs:135); each ring slot's stream needs a hardware queue of
# its own to overlap with the others, and the runtime's default is 4.  Must be set before the HIP runtime starts.
# N > 1: 12 launches of 4 frames each in flight on 16 hardware queues -- the exchange stream and RCCL's streams get queues of their own, and
# the command processor's cliff at 24 queues in use (3x slower: profiles/README.md r1k) stays far away.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between the processes of a job on this driver


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--lights", type=int, default=1, choices=[1, 4])
    ap.add_argument("--scene", default="sponza", choices=["sponza", "bistro"], help="sponza = BASELINE configs 2/3/5 (the default is config 2); bistro = config 4 (2.8 M triangles, one directional light)")
    ap.add_argument("--detail", type=float, default=1.0, help="scene detail (1.0 = the 262k-triangle config)")
    ap.add_argument("--glb", default=None, help="a .glb that satisfies the reference's reader (gltf_model_reader.rs:62-63, :643-681: one mesh, one buffer, tangents, albedo + ORM + normal "
                                                "textures) instead of the synthetic scene: loaded through art_scene_add_glb, set up like main.rs:23-66; the line then says data: \"real glb\". "
                                                "Default: assets/*.glb if one is there (SURVEY.md 8d), else the synthetic scene")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="ring of per-frame streams / buffers like the reference's FrameData ring (renderer.rs:135, which keeps 3); "
                                                                      "default on one GPU: 8 for every run (round 3 took 3 for runs of at most 32 steps: a still camera's 20-frame burst is 3 %% faster through 3 slots -- 0.164 against 0.169 ms a frame -- "
                                                                      "but a camera or a model that moves costs 7-25 %% there and 1-3 %% here: tools/camera_leg_probe.py, profiles/README.md round 4; 1 000 steps: 19 650 with 8, 19 350 with 16, 18 900 with 3); "
                                                                      "12 launches when the frame is sharded")
    ap.add_argument("--ao", type=int, default=0, help="BASELINE config 5: N ray-traced AO rays per hit pixel after each frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--watchdog-seconds", type=float, default=-1.0, help="if the run has not finished after this long, every rank prints its Python stack and exits 1 (0 = off; default: off "
                    "for one GPU, 300 for N > 1 -- below the 600 s after which the driver kills a bench run -- whose exchange no machine in reach could rehearse over RCCL): "
                    "a hung job then ends with a diagnosis instead of holding the GPU until someone kills it")
    ap.add_argument("--settle-seconds", type=float, default=1.0, help="untimed set-up before the warm-up: frames are traced for at least this long (and at least 3 ring depths), so the wave plan "
                    "has settled and the GPU has left its idle clocks; 0 = the 3 ring depths only")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall clock the CPU baseline repeats the frame for (the contract: a bounded sample, ~10-30 s of CPU work)")
    ap.add_argument("--plain", action="store_true", help="only the contract's timed region (profiling passes: no single-frame spans, no steady-state / camera-path legs, no CPU baseline)")
    ap.add_argument("--tuning", default="", help="A/B sweeps: ArtTuning fields for the benchmarked context, key=value[,key=value...] (include/art.h: frame_form, tree_builder, frame_waves, "
                                                 "block_order, split_alpha ...); the default -- none -- is the product")
    ap.add_argument("--camera-walk", type=int, default=64, help="N = 1 extra leg `camera_path`: the camera moves every frame the way the reference's does (main.rs:69-131: a key held down, a hand on the "
                                                                "mouse) along a closed loop of this many poses (0 = skip); ray counts of every eighth pose are checked against the oracle's committed ones")
    ap.add_argument("--camera-path", type=int, default=8, help="N = 1 extra leg `camera_jumps`: the camera JUMPS every frame along a closed path of this many poses 0.2 units / 15 degrees apart (0 = skip); ray counts of every pose "
                                                               "are checked against the oracle's committed ones")
    ap.add_argument("--moving-model", type=int, default=64, help="N = 1 extra leg: one model of the scene (its last primitive) is moved and rotated before every frame along a closed path of this many "
                                                                "poses (art_scene_set_model_matrix: a device refit in front of each frame, VkModel::set_model_matrix + the reference's per-frame TLAS); 0 = skip")
    ap.add_argument("--gather-launches", type=int, default=-1, help="N>1: ring slots (launches) per exchange; 0 = the whole ring (the slots are contiguous, so a group travels as one message "
                    "per peer); default: a quarter of the timed launches, at most the ring -- a run of a few launches then still overlaps its exchanges with its tracing "
                    "instead of paying one exchange of everything behind the last frame")
    ap.add_argument("--second-placement-seconds", type=float, default=120.0, help="N>1: the second of the two placements runs under a timer of this many seconds; past it rank 0 prints the first "
                    "placement's line and every rank exits with code 0 (0 = no timer)")
    ap.add_argument("--rehearse-hang", action="store_true", help="N>1, --backend gloo: the second placement's first exchange never returns (a test of --second-placement-seconds)")
    ap.add_argument("--one-placement", action="store_true", help="N>1: time only the placement --roots names (default: both, back to back -- `value` is --roots', the other one's rate is reported beside it)")
    ap.add_argument("--roots", default="spread", choices=["spread", "rank0"],
                    help="N>1: where frames are assembled (the placement `value` is measured with; the other is timed too and reported as value_rank0_root / value_spread_roots).  'spread' (default) = frame f on rank f mod N: every group of launches is one grouped ncclSend / ncclRecv (each frame's "
                         "gather, all at once), every xGMI link carries its share in both directions; 'rank0' = one ncclGather per group to rank 0, which then receives over its own "
                         "links only (2 GPUs: 12.4 MB of RGB32F tiles per 1080p frame over ONE link).  A dedicated compositor implies rank0")
    ap.add_argument("--compositor", default="shared", choices=["dedicated", "shared"],
                    help="N>1: 'shared' = rank 0 traces a share AND receives / un-tiles every frame; 'dedicated' = rank 0 only composites, ranks 1..N-1 trace 1/(N-1) each "
                         "(rehearsed on one GPU: a root that also traces a 1/8 share spends 35.5 us per frame where the tracers of a 7 + 1 layout need 39.1, profiles/README.md r1n)")
    ap.add_argument("--frames-per-launch", type=int, default=0, help="N>1: frames one launch traces (1..4; default: the largest of 4, 2, 1 that divides --steps and --warmup's launches; "
                                                                       "a launch costs ~7 us of machine time whatever it traces, which a 1/8 share feels)")
    ap.add_argument("--root-relief", type=int, default=-1, help="N>1, shared compositor: 1/256ths of rank 0's share handed to the other ranks (default 8 per GPU)")
    ap.add_argument("--gather", default="fp32", choices=["fp32", "packed"], help="N>1 exchange payload: RGB32F tiles -- the HDR buffer, 12 B per pixel: its alpha is the constant 1 and stays at home (default, the contract) -- or "
                                                                                  "B10G11R11_UFLOAT_PACK32 words, the reference's colour image format (renderer.rs:268), 4 B per pixel")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL inside libart (ncclGather on the tile buffers); gloo = rehearsal of the same loop with the "
                                                                                  "collective replaced by a host function (tiles staged through host memory, ranks may share one GPU)")
    return ap.parse_args()


def host_cores():
    n = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)))
    try:  # a cgroup CPU quota (e.g. 16 CPUs of a 256-thread host) is the real core budget
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def spawn_ranks(args):
    """`python bench.py --gpus N` started plainly (no launcher, no WORLD_SIZE): this process -- which has not touched the GPU and never will -- starts the N ranks the way the
    driver does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`) as a CHILD process, relays its output
    (rank 0's line is the job's) and leaves with its exit code.  (Never an exec: a process that initialised the GPU must not be replaced, and this one must not initialise it.)"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without a launcher: starting the ranks as a child process: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    watchdog = args.watchdog_seconds if args.watchdog_seconds >= 0 else (300.0 if int(os.environ.get("WORLD_SIZE", "1")) > 1 else 0.0)
    if watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(watchdog, exit=True)
    import numpy as np
    import torch
    import torch.distributed as dist
    from araytracingjourney_amd import renderer, scenes
    import roofline as RL

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or plainly: bench.py starts its own ranks)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libart has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()   # rehearsal: ranks may share one GPU
    torch.cuda.set_device(local_rank)
    if world > 1:
        # control plane (the job's id, barriers, the max over ranks of the wall clock) on gloo; the DATA path -- the gather of the tiles --
        # is RCCL inside libart (art_mgpu_*), which needs nothing from torch
        dist.init_process_group("gloo")

    # ---- the scene -------------------------------------------------------------------------------------------------------------------
    W, H = args.width, args.height
    glb = args.glb
    if glb is None and os.path.isdir(os.path.join(ROOT, "assets")):
        found = sorted(f for f in os.listdir(os.path.join(ROOT, "assets")) if f.lower().endswith(".glb"))
        glb = os.path.join(ROOT, "assets", found[0]) if found else None
    if glb:
        sc = scenes.from_glb(glb, lights=scenes.sponza_lights(args.lights))   # the oracle's copy of what the C++ reader yields
        lights = sc.lights
    elif args.scene == "bistro":
        sc = scenes.bistro_like(args.detail)
        lights = sc.lights
    else:
        sc = scenes.sponza_like(args.detail)
        lights = scenes.sponza_lights(args.lights)
        sc = scenes.Scene(sc.name, sc.primitives, sc.camera, lights)
    F = max(1, min(22, args.frames_in_flight)) if args.frames_in_flight > 0 else (12 if world > 1 else 8)   # ONE depth for every one-GPU run, the driver's 20 steps and the long runs and the counter passes alike
    packed = world > 1 and args.gather == "packed"
    dedicated = world > 1 and args.compositor == "dedicated"
    G = world - 1 if dedicated else world           # shards of the frame = ranks that trace
    renders = not (dedicated and rank == 0)
    # rank 0 also receives and un-tiles every frame: its share shrinks by 1/32 per GPU (2 GPUs: 6 %, 8 GPUs: 25 % of an equal share), which
    # is what levels its loop with the others' on the rehearsal (profiles/README.md r1n: 35.7 -> 31 us per frame at N = 8)
    shard = renderer.mgpu_shard(rank, world, dedicated) if world > 1 else (0, 1)

    tuning = {k: (float(v) if k in ("split_alpha", "refit_rebuild_ratio") else int(v)) for k, v in (kv.split("=") for kv in args.tuning.split(",") if kv)} or None

    def make_renderer(**kw):
        if glb:   # through the real ingest: art_glb_open + art_scene_add_glb (renderer.rs:346)
            from araytracingjourney_amd import model_reader as mr
            r_ = renderer.Renderer((W, H), **kw)
            r_.add_model_glb(mr.GltfModelReader(glb, True, mr.COERCE_B8G8R8A8), scenes.scale_matrix(2.0))
            cam = r_.camera_mut()
            cam.set_pos(sc.camera["pos"]); cam.set_dir(sc.camera["dir"]); cam.set_fovy(sc.camera["fovy"]); cam.set_znear(sc.camera["znear"]); cam.set_zfar(sc.camera["zfar"])
            for d in lights:
                r_.lights_mut().push_dict(d)
            r_.prepare_first_frame()
            return r_
        return renderer.renderer_for_scene(sc, (W, H), **kw)

    def run_job(roots):               # one whole job: context (+ the sharded frame behind the C ABI), settle, warm-up, the timed region.  roots: None (one GPU) | "spread" | "rank0"
        spread = roots == "spread"
        relief = 0 if (world == 1 or dedicated or spread) else min(255, (8 * world if args.root_relief < 0 else args.root_relief))
        r = make_renderer(device=local_rank, shard=shard, frames_in_flight=F, packed_tiles=packed, tuning=tuning, root_relief=relief)
        B = 1                             # frames per launch
        if (world > 1 or args.frames_per_launch > 1) and not args.ao:
            B = args.frames_per_launch if args.frames_per_launch > 0 else next(b for b in (4, 2, 1) if args.steps % b == 0)
            if args.steps % B:
                raise SystemExit(f"--steps {args.steps} is not a multiple of --frames-per-launch {B}")
            r.set_frames_per_launch(B)
        r.upload_state()

        # launches per exchange: explicit, or a quarter of the timed launches (libart rounds it down to a divisor of the ring)
        gather_launches = args.gather_launches if args.gather_launches >= 0 else max(1, min(F, (args.steps // B) // 4))
        # ---- N > 1: the sharded frame behind the C ABI -------------------------------------------------------------------------------------
        mg, transport = None, None
        if world > 1:
            if args.backend == "nccl":
                ids = [renderer.mgpu_unique_id() if rank == 0 else None]   # (a communicator of its own for every placement)
                dist.broadcast_object_list(ids, src=0)
                mg = renderer.MultiGpu(r, rank, world, unique_id=ids[0], dedicated=dedicated, launches_per_gather=gather_launches, spread=spread)
                transport = "RCCL inside libart (art_mgpu_*): " + ("grouped ncclSend / ncclRecv" if spread else "ncclGather")
            else:
                import ctypes as C
                hip = C.CDLL("libamdhip64.so")
                hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

                def gloo_gather(send, nbytes, recv, root, stream):   # the collective of the rehearsal: same call sequence, payload through host memory
                    if args.rehearse_hang and jobs:
                        time.sleep(1e6)
                    host = torch.empty(nbytes, dtype=torch.uint8)
                    assert hip.hipMemcpy(host.data_ptr(), send, nbytes, 2) == 0
                    parts = [torch.empty_like(host) for _ in range(world)] if rank == root else None
                    dist.gather(host, parts, dst=root)
                    if rank == root:
                        for w_, part in enumerate(parts):
                            assert hip.hipMemcpy(recv + w_ * nbytes, part.data_ptr(), nbytes, 1) == 0
                mg = renderer.MultiGpu(r, rank, world, dedicated=dedicated, launches_per_gather=gather_launches, exchange=gloo_gather, spread=spread)
                transport = "host function over gloo (rehearsal)"

        def control_plane(what):          # no collective of the control plane while exchanges of the data path are queued (they are submitted lazily): ranks would meet in different collectives
            if mg:
                mg.assert_quiescent(what)

        def step():                       # one launch: B frames of this rank's share (+ its part of the exchange)
            if mg:
                mg.trace()
            else:
                r.trace()
            if args.ao and renders:
                r.trace_ao(args.ao)       # per tile from the local G-buffer: no extra exchange

        def fence():
            if mg:
                mg.flush()                # every frame traced so far is on the root, un-tiled
            r.sync()
            torch.cuda.synchronize()
            if world > 1:                 # the contract's bracket: barrier + synchronize (one process: the synchronize above is the bracket)
                control_plane("the fence's barrier")
                dist.barrier()
                torch.cuda.synchronize()

        # ---- settle (untimed set-up, like the scene build): the wave plan has sampled frames and re-planned, every stream has run -----------------
        # ... and the GPU has left its idle clocks (tools/fenced_timeline.py).  Every rank makes the SAME number of launches (each is part of a gather):
        # they come in rounds of 3 ring depths, and rank 0's clock decides after each round whether another one follows.
        settle_launches, t_settle = 0, time.perf_counter()
        while True:
            for _ in range(3 * F):
                step()
            settle_launches += 3 * F
            more = [time.perf_counter() - t_settle < args.settle_seconds]
            if world > 1:
                fence()       # every gather of the round is through (they are submitted lazily, from later launches or the flush) before anyone waits on the control plane
                control_plane("the settle round's broadcast")
                dist.broadcast_object_list(more, src=0)
            if not more[0]:
                break
        fence()
        # ---- the contract: W untimed warm-up steps, then EXACTLY K steps between two fences ------------------------------------------------------
        for _ in range((args.warmup + B - 1) // B):   # a step() is one launch = B frames
            step()
        fence()
        r.collect_timings()  # drop the warm-up frames from the per-stage event sums
        t0 = time.perf_counter()
        for _ in range(args.steps // B):
            step()
        fence()
        wall = time.perf_counter() - t0
        stage, n_timed = r.collect_timings()  # HIP events on the frames' own streams, over the timed frames (last <= 128)

        # ---- outside the timed region: the gathered frame must equal an unsharded render of the same frame, bit for bit ---------------------------
        frame_ok = None
        if world > 1 and (rank == 0 or spread):          # every rank that assembles frames checks the newest one it holds
            got = mg.read_frame()
            whole = make_renderer(device=local_rank)
            whole.render_frame()
            if packed:   # the assembled frame is the packed colour image: against the single GPU's (art_present packs it, vk_rt_lightning_shadows.rs:152)
                whole.present()
                frame_ok = bool(np.array_equal(got, whole.read_packed()[0]))
            else:
                frame_ok = bool(np.array_equal(got.view(np.uint32), whole.read_color().view(np.uint32)))
            whole.close()
        if world > 1 and spread:
            control_plane("the frame check's all_gather")
            oks = [None] * world
            dist.all_gather_object(oks, frame_ok)
            frame_ok = all(oks)
        counts = mg.counts() if mg else None
        st = r.stats()
        if not renders:                   # the compositor traced nothing
            st = dict(st, primary_rays=0, shadow_rays=0, ao_rays=0, hit_pixels=0)
        rays_local = st["primary_rays"] + st["shadow_rays"] + st["ao_rays"]
        if world > 1:
            t = torch.tensor([wall, float(rays_local), float(st["shadow_rays"]), stage["frame_ms"]], dtype=torch.float64)
            tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            wall = float(tmax[0])
            rays_total, shadow_total = float(tsum[1]), float(tsum[2])
            stage = dict(stage, frame_ms=float(tmax[3]))
        else:
            rays_total, shadow_total = float(rays_local), float(st["shadow_rays"])
        if mg:
            mg.close()
        if world > 1:                     # the next placement's context gets this one's streams back (24 hardware queues in use is a 3x cliff, profiles/README.md r1k)
            r.close()
            r = None
        return dict(r=r, wall=wall, stage=stage, n_timed=n_timed, frame_ok=frame_ok, counts=counts, st=st, rays_total=rays_total, shadow_total=shadow_total, B=B, transport=transport,
                    relief=relief, spread=spread, settle_launches=settle_launches, step=step, fence=fence, value=rays_total / (wall / args.steps) / 1e6, ms_per_step=wall * 1e3 / args.steps)

    if world > 1:
        placements = [args.roots] if (dedicated or args.one_placement) else [args.roots, "rank0" if args.roots == "spread" else "spread"]
        if dedicated:
            placements = ["rank0"]
    else:
        placements = [None]
    def finish(main_pl, other, bailing=False):   # everything behind the timed region(s): the extra legs (one GPU), the roofline, the line
        main_job = jobs[main_pl]
        r, wall, stage, n_timed, frame_ok, counts, st, rays_total, shadow_total, B, transport, relief, spread, settle_launches, step, fence = (main_job[k] for k in (
            "r", "wall", "stage", "n_timed", "frame_ok", "counts", "st", "rays_total", "shadow_total", "B", "transport", "relief", "spread", "settle_launches", "step", "fence"))
        mg = None

        extras = world == 1 and not args.plain
        # ---- steady state: the same K frames with the ring kept full on both sides (device timestamps behind the last priming frame and behind frame K) ----
        steady = None
        if extras and not args.ao and args.steps >= 4 * F:   # (a run of fewer than four ring depths is all fill and drain: nothing steady to report)
            for _ in range(2 * F):
                r.trace()
            r.timestamp_mark(0)
            for _ in range(args.steps):
                r.trace()
            r.timestamp_mark(1)
            for _ in range(F):            # frames behind the mark: the ones in front of it never run on an emptying GPU
                r.trace()
            ms = r.timestamp_elapsed_ms()
            fence()
            steady = dict(ms_per_step=ms / args.steps, frames=args.steps, protocol=f"{2 * F} priming frames, device timestamp behind the last of them, {args.steps} frames, timestamp, {F} more frames; no host "
                                                                                   "synchronisation in between: what a render loop sees, where `value` (fenced on both sides) also pays the fill and the drain of the ring")
        # ---- one frame on the GPU at a time, each timed by its own HIP events (median / p10 / p90 over 100 frames) -----------------
        alone = None
        if extras:
            spans = []
            for _ in range(100):
                step()
                fence()
                spans.append(r.collect_timings()[0]["frame_ms"])
            spans.sort()
            alone = dict(median_ms=spans[50], p10_ms=spans[10], p90_ms=spans[90], frames=100)
            if not args.ao and F > 1:
                # the same frame in a context that keeps ONE frame in flight: there the wave plan splits the blocks whose packets crawl (with 16 in
                # flight nothing needs splitting), which is what a caller that wants a frame's latency rather than frames per second would use
                one = make_renderer(device=local_rank, frames_in_flight=1)
                one.upload_state()
                for _ in range(24):               # the plan settles within a few frames
                    one.trace(); one.sync()
                one.collect_timings()
                spans = []
                for _ in range(60):
                    one.trace(); one.sync()
                    spans.append(one.collect_timings()[0]["frame_ms"])
                spans.sort()
                alone["single_frame_context"] = dict(median_ms=spans[30], p10_ms=spans[6], p90_ms=spans[54], frames=60, split_blocks=one.stats()["split_blocks"])
                one.close()

        # ---- a moving camera: every frame a different pose (no frame in flight shares its BVH path with its neighbours) --------------------------
        campath = None
        tag = {(1920, 1080, 1): "c2_sponza_like_1080p_1light", (3840, 2160, 4): "c3_sponza_like_2160p_4lights"}.get((W, H, args.lights)) if (args.scene == "sponza" and args.detail == 1.0 and not glb) else None
        if args.scene == "bistro" and args.detail == 1.0 and (W, H) == (1920, 1080) and not glb:
            tag = "c4_bistro_like_1080p_1light"
        def camera_leg(poses, gold, every, protocol):
            cams = [renderer.Camera(p["pos"], p["dir"], W / H, p["fovy"], p["znear"], p["zfar"]) for p in poses]
            rays_pose, checked = [], 0
            for i, cam in enumerate(cams):            # the poses once alone: their ray counts, against the oracle's committed ones (every `every`-th pose has one)
                if i % every:
                    rays_pose.append(None)
                    continue
                r._camera = cam
                r.upload_state(); r.trace(); r.sync()
                ps = r.stats()
                if gold:
                    g = gold["poses"][i // every]
                    assert (ps["shadow_rays"], ps["hit_pixels"]) == (g["shadow_rays"], g["hit_pixels"]), f"camera pose {i}: GPU ray counts differ from the oracle's"
                    checked += 1
                rays_pose.append(ps["primary_rays"] + ps["shadow_rays"])
            known = [x for x in rays_pose if x is not None]
            rays_pose = [x if x is not None else sum(known) / len(known) for x in rays_pose]   # (poses in between: the mean of the counted ones -- they differ by a thousandth)

            at = [0]                            # the motion goes on across the settle, warm-up and timed parts (no jump back to pose 0 in between)

            def moving(n):
                for _ in range(n):
                    r._camera = cams[at[0] % len(cams)]
                    at[0] += 1
                    r.upload_state()
                    r.trace()
            t_settle = time.perf_counter()      # the leg settles like the timed region itself: its own frames for --settle-seconds (the legs in front of it left the GPU idling between
            while True:                         # single fenced frames: twenty frames on idle clocks measure the clocks), then W warm-up frames and a fence
                moving(3 * F)
                if time.perf_counter() - t_settle >= args.settle_seconds:
                    break
            fence()
            moving(args.warmup)
            fence()
            first = at[0]
            c0 = time.perf_counter()
            moving(args.steps)
            fence()
            cwall = time.perf_counter() - c0
            rays_moved = sum(rays_pose[(first + i) % len(cams)] for i in range(args.steps))
            return dict(poses=len(cams), value=rays_moved / cwall / 1e6, unit="Mray/s", ms_per_step=cwall * 1e3 / args.steps, rays_per_frame_min=min(rays_pose), rays_per_frame_max=max(rays_pose),
                        ray_counts_checked_against_oracle=checked, protocol=protocol)

        def gold_of(name):
            path = os.path.join(ROOT, "tests", "golden", f"{tag}.{name}.json") if tag else None
            return json.load(open(path)) if path and os.path.exists(path) else None
        camjumps = None
        if extras and not args.ao and args.camera_walk > 0:
            # camera_path: the reference's own camera motion (main.rs:69-131) -- a key held down and a hand on the mouse, at the frame rate this library renders at
            campath = camera_leg(scenes.camera_walk(sc, args.camera_walk), gold_of(f"camera_walk_{args.camera_walk}"), 8,
                                 "art_set_camera before every frame; settled, warmed up and fenced on both sides like `value`; the camera moves as the reference's does (main.rs:80-124: 0.002 units per ms of frame time along the view "
                                 "direction, 0.002 rad per mouse count at 1 000 counts / s) at 6 000 frames / s -- 0.00033 units and 0.00033 rad a frame, a closed loop out and back")
        if extras and not args.ao and args.camera_path > 0:
            # camera_jumps: a pose 0.2 units and 15 degrees from the last one EVERY frame (round 1-3's camera_path): no frame in flight shares its heavy blocks with its neighbours -- a stress
            # test of what depends on frame-to-frame coherence (the wave plan), not something a render loop does
            camjumps = camera_leg(scenes.camera_path(sc, args.camera_path), gold_of(f"camera_path_{args.camera_path}"), 1,
                                  "art_set_camera before every frame with a pose 0.2 units / 15 degrees away from the last (a closed path of 8 poses); settled, warmed up and fenced on both sides like `value`: a stress leg -- "
                                  "nothing a frame learns about its heavy blocks holds for the next")
        if campath or camjumps:
            r._camera = renderer.Camera(sc.camera["pos"], sc.camera["dir"], W / H, sc.camera["fovy"], sc.camera["znear"], sc.camera["zfar"])
            r.upload_state()

        # ---- a moving model: art_scene_set_model_matrix before every frame (row a3: the reference rebuilds its TLAS every frame so that models can move) --------
        moving_model = None
        if extras and not args.ao and not glb and args.moving_model > 0 and len(sc.primitives) > 1:
            import math
            r.close()                     # the benchmarked context is done (its streams go back to the pool: two rings of 16 would share the 16 hardware queues)
            mv = renderer.Renderer((W, H), device=local_rank, frames_in_flight=F, tuning=tuning, dynamic_scene=True)   # (the host says its models will move: the ring of structure versions is made by the build)
            mv.add_model(sc.primitives[:-1])
            mv.add_model(sc.primitives[-1:])
            cam = mv.camera_mut()
            cam.set_pos(sc.camera["pos"]); cam.set_dir(sc.camera["dir"]); cam.set_fovy(sc.camera["fovy"]); cam.set_znear(sc.camera["znear"]); cam.set_zfar(sc.camera["zfar"])
            for d in lights:
                mv.lights_mut().push_dict(d)
            mv.prepare_first_frame()
            mv.upload_state()
            base = np.vstack([np.asarray(sc.primitives[-1].model, np.float64).reshape(3, 4), [0, 0, 0, 1]])
            poses = []
            for i in range(args.moving_model):   # a closed loop, out and back: the model is carried along x and turned about y at 2 units / s and 2 rad / s -- the speeds the reference's own
                k = min(i, args.moving_model - i)  # controls move things at (main.rs:80-124) -- at the 6 000 frames / s this library renders the scene at: 0.00033 units and 0.00033 rad a frame
                a = 2.0 / 6000.0 * k
                ry = np.array([[math.cos(a), 0, math.sin(a), 0], [0, 1, 0, 0], [-math.sin(a), 0, math.cos(a), 0], [0, 0, 0, 1]])
                t = np.eye(4); t[:3, 3] = (2.0 / 6000.0 * k, 0.0, 0.0)
                poses.append(np.ascontiguousarray((t @ ry @ base)[:3], np.float32))
            model = mv.models_mut()[1]
            rays_pose, refit_alone = [], []
            for m in poses:                       # every pose once alone: its ray count, and the refit's device time with nothing else on the GPU
                model.set_model_matrix(m)
                mv.trace(); mv.sync()
                ps = mv.stats()
                rays_pose.append(ps["primary_rays"] + ps["shadow_rays"]); refit_alone.append(ps["refit_ms"])

            mat = [0]

            def moved(n):
                for _ in range(n):
                    model.set_model_matrix(poses[mat[0] % len(poses)])
                    mat[0] += 1
                    mv.trace()
            t_settle = time.perf_counter()      # (settled like the timed region: see the camera legs)
            while True:
                moved(3 * F)
                if time.perf_counter() - t_settle >= args.settle_seconds:
                    break
            mv.sync()
            moved(args.warmup); mv.sync()
            mfirst = mat[0]
            m0 = time.perf_counter()
            moved(args.steps); mv.sync()
            mwall = time.perf_counter() - m0
            ms_ = mv.stats()
            refit_alone.sort()
            # the same model leaves and re-enters the structure (the residency rule, vk_model.rs:334-345): art_scene_set_primitive_enabled on a built scene is a refit too
            from araytracingjourney_amd._lib import check
            residency = []
            for on in (0, 1, 0, 1):
                for pid in model.primitive_ids:
                    check(mv._L.art_scene_set_primitive_enabled(mv._ctx, pid, on))
                mv.trace(); mv.sync()
                residency.append(mv.stats()["refit_ms"])
            assert not mv.needs_build() and mv.stats()["rebuilds"] == ms_["rebuilds"]
            moving_model = dict(poses=len(poses), value=sum(rays_pose[(mfirst + i) % len(poses)] for i in range(args.steps)) / mwall / 1e6, unit="Mray/s", ms_per_step=mwall * 1e3 / args.steps,
                                refit_ms=refit_alone[len(refit_alone) // 2], refit_ms_max=refit_alone[-1], refits=ms_["refits"], rebuilds=ms_["rebuilds"], refit_cost_ratio=ms_["refit_cost_ratio"],
                                moving_triangles=sc.primitives[-1].n_tris, build_ms=ms_["build_ms"], residency_change_ms=sorted(residency)[len(residency) // 2],
                                first_move_ms=ms_["first_move_ms"], versions_ms=ms_["versions_ms"],
                                protocol="art_scene_set_model_matrix before every frame (one model = the scene's last primitive, 62 % of its triangles, carried and turned at 2 units / s and 2 rad / s -- 0.00033 a frame -- out and back); settled, warmed up and fenced on both sides like `value`; "
                                         "versions_ms = host time of making the ring of structure versions (inside art_scene_build: ART_FLAG_DYNAMIC_SCENE; first_move_ms = what the first moved frame paid, 0 then); "
                                         "refit_ms = device time of one refit (all triangle records + every 4-wide node) with nothing else on the GPU, median over the poses; "
                                         "residency_change_ms = the same for the model leaving / re-entering the structure (art_scene_set_primitive_enabled: no build)")
            mv.close()

        if rank != 0:
            if world > 1 and not bailing:
                dist.destroy_process_group()
            return

        ms_per_step = wall * 1e3 / args.steps
        value = rays_total / (wall / args.steps) / 1e6

        # ---- the oracle's counters for this exact frame (canonical LBVH: per ray = SURVEY.md 8d's contract figure, per 8x8 packet) ------------------
        fx = os.path.join(ROOT, "tests", "golden", f"{tag}.stats.json") if tag and not args.ao else None
        ost = json.load(open(fx)) if fx and os.path.exists(fx) else None

        # ---- CPU baseline: the scalar C oracle on this host's cores, on a bounded sample of the same frame ---------------------------------------
        cpu = None
        if not args.no_cpu_baseline and not args.plain and not args.ao and world == 1:   # rank 0 at N = 1 only (the contract)
            from oracle import orc
            ncores = host_cores()
            S = orc.Scene(sc.primitives, morton_bits=30)
            cam = orc.camera_from_params(sc.camera["pos"], sc.camera["dir"], W / H, sc.camera["fovy"], sc.camera["znear"], sc.camera["zfar"])
            L = orc.make_lights(lights)
            S.render(cam, L, len(lights), W, H, 0, H, threads=ncores, reuse=True)       # warm-up frame
            reps, cdt, cst, frame_s = 0, 0.0, None, []
            c0 = time.perf_counter()
            while (cdt < args.cpu_seconds or reps < 3) and reps < 400:   # the same whole frame, repeated for ~10 s of wall clock (three frames at least: BASELINE.md 3 asks the median of 3)
                f0 = time.perf_counter()
                cst = S.render(cam, L, len(lights), W, H, 0, H, threads=ncores, reuse=True)["stats"]
                frame_s.append(time.perf_counter() - f0)
                reps += 1
                cdt = time.perf_counter() - c0
            rays1 = cst["primary_rays"] + cst["shadow_rays"]
            frame_s.sort()
            med = frame_s[len(frame_s) // 2]
            c1 = time.perf_counter()
            st1 = S.render(cam, L, len(lights), W, H, H // 2 - 128, H // 2 + 128, threads=1, reuse=True)["stats"]   # one thread, a 256-row band
            dt1 = time.perf_counter() - c1
            cpu = dict(value=rays1 / med / 1e6, unit="Mray/s", cores=ncores, kind="port", frames=reps, value_mean=rays1 * reps / cdt / 1e6,
                       value_1thread=(st1["primary_rays"] + st1["shadow_rays"]) / dt1 / 1e6,
                       sample=f"the median of {reps} renderings of the same {W}x{H} frame ({rays1} rays each, {cdt:.1f} s wall, one warm-up before them) on {ncores} threads; 1-thread figure "
                              f"on rows [{H // 2 - 128},{H // 2 + 128}) ({dt1:.1f} s); scalar C oracle (stands in for the scalar Rust tracer: no Rust "
                              "toolchain in this image), threads over 32x32-pixel tiles")
            if ost is None:     # no committed counters for this workload (other extents, a .glb): the oracle's, counted now
                pk, _ = orc.packet_stats(S, cam, L, len(lights), W, H, threads=ncores)
                ost = dict(cst, **pk)
        if ost is not None and world == 1:
            assert ost["shadow_rays"] == st["shadow_rays"] and ost["hit_pixels"] == st["hit_pixels"], \
                f"GPU ray counts differ from the oracle's: {st['shadow_rays']}/{st['hit_pixels']} vs {ost['shadow_rays']}/{ost['hit_pixels']}"

        # ---- roofline (tools/roofline.py re-derives every number below from profiles/) -------------------------------------------------------------
        # Which roof binds a launch is a MEASURED statement: instruction and traffic counts come from a committed rocprofv3 --pmc pass of this very workload
        # (profiles/current_pmc.json, keyed by workload: PMC counters cannot be read from inside the benchmarked process).  Without one -- another extent,
        # a .glb, N > 1 -- the line carries the algorithmic fractions only and says bound: null; it never guesses "hbm".
        roof = None
        workload = workload_name(sc, glb, W, H, lights, shadow_total, args)
        cur_path = os.path.join(ROOT, "profiles", "current_pmc.json")
        cur = json.load(open(cur_path)) if os.path.exists(cur_path) else {}
        entry = (cur.get("workloads") or {}).get(workload) if world == 1 else None
        pmc = entry["pmc"] if entry else {}
        if args.ao and world == 1 and st.get("frame_launches") == 1:
            # config 5: the dominant kernel is the AO launch's persistent tracer.  Its share of the machine's time per step = the step minus the same frames without
            # their AO pass, timed here the same way (fenced on both sides)
            for _ in range(2 * F):
                r.trace()
            fence()
            f0 = time.perf_counter()
            for _ in range(args.steps):
                r.trace()
            fence()
            frame_only_ms = (time.perf_counter() - f0) * 1e3 / args.steps
            us = max(ms_per_step - frame_only_ms, 1e-3) * 1e3
            gold_ao = os.path.join(ROOT, "tests", "golden", "c5_sponza_like_2160p_16spp_ao.stats.json")
            ga = json.load(open(gold_ao)) if (args.scene == "sponza" and (W, H) == (3840, 2160) and args.detail == 1.0 and not glb and args.ao == 16 and os.path.exists(gold_ao)) else None
            ab = None
            if ga and ga["ao_rays"] == st["ao_rays"]:
                ab = dict(contract=(32 + 1) * ga["ao_rays"] + 64 * ga["n_int_ao"] + 48 * ga["n_tri_ao"])   # SURVEY 8(d) per ray on the canonical LBVH: ray + nodes + triangles + the occlusion byte
            fr = RL.fractions(pmc, us, ab)
            roof = dict(bound="valu_issue" if "valu_issue_frac" in fr else None, kernel="k_trace_ao (the AO launch's persistent per-ray tracer: rays made by the whole wave into a pool in LDS)",
                        achieved=pmc["SQ_INSTS_VALU"] / (us * 1e-6) / 1e9 if "SQ_INSTS_VALU" in pmc else None, peak=RL.SIMDS * RL.CLOCK_HZ / RL.VALU_CYCLES / 1e9, unit="G wave-instructions/s",
                        frac=fr.get("valu_issue_frac"), traffic=fr.get("hbm_bytes_per_launch"), hbm_frac=fr.get("hbm_frac"), valu_issue_frac=fr.get("valu_issue_frac"), salu_issue_frac=fr.get("salu_issue_frac"),
                        valu_lane_utilisation=fr.get("valu_lane_utilisation"),
                        useful_valu_frac=(fr["valu_issue_frac"] * fr["valu_lane_utilisation"]) if ("valu_issue_frac" in fr and "valu_lane_utilisation" in fr) else None,
                        contract_frac=fr.get("contract_frac"), l2_hit_rate=fr.get("l2_hit_rate"), machine_us_per_launch=us, frame_only_ms_per_step=frame_only_ms, ao_rays_per_launch=st["ao_rays"],
                        algorithmic_bytes_per_launch=ab, pmc_source=entry["source"] if entry else None, pmc_kernel_source_sha16=entry.get("kernel_source_sha16") if entry else None,
                        pmc_stale=(entry.get("kernel_source_sha16") != RL.kernel_source_hash()) if entry else None,
                        note="the AO launch: machine time = this run's ms_per_step minus the same frames without their AO pass; frac = vector wave-instructions of the launch (committed rocprofv3 --pmc "
                             "pass) x 2 cycles over what 1 024 SIMDs issue in that time; useful_valu_frac = frac x the share of lanes active per vector instruction (incoherent rays: a wave's "
                             "lanes finish and wait at different times); contract_frac = SURVEY.md 8(d)'s per-ray bytes on the canonical LBVH / 8 TB/s (cache-resident: NOT an achieved bandwidth)")
        elif ost is not None and st.get("frame_launches") == 1:
            ab = RL.algorithmic_bytes(ost, len(lights))
            us = ms_per_step * 1e3 * world                     # machine time one GPU spends per frame: each rank's launch handles 1/world of the frame's rays
            share = 1.0 / world
            ab_launch = {k: v * share * B for k, v in ab.items()}   # what ONE launch (B frames of a 1/world share) accounts for
            fr = RL.fractions(pmc, us * B, ab_launch)
            kernel_ms = stage["primary_ms"]                    # HIP events on the launch's own stream: one launch's span, overlapped by the others in flight
            binding = max((k for k in ("valu_issue_frac", "salu_issue_frac", "hbm_frac") if k in fr), key=lambda k: fr[k], default=None)
            if binding is None:
                roof = dict(bound=None, kernel="k_frame", achieved=None, peak=None, unit=None, frac=None)   # no counters for this workload: nothing measured says which roof binds it
            elif binding == "hbm_frac":
                roof = dict(bound="hbm", kernel="k_frame", achieved=fr["hbm_bytes_per_launch"] / (us * B * 1e-6) / 1e9, peak=RL.HBM_PEAK / 1e9, unit="GB/s", frac=fr["hbm_frac"])
            else:
                per_s = {"valu_issue_frac": RL.SIMDS * RL.CLOCK_HZ / RL.VALU_CYCLES, "salu_issue_frac": RL.CUS * RL.CLOCK_HZ}[binding]
                n_inst = pmc["SQ_INSTS_VALU" if binding == "valu_issue_frac" else "SQ_INSTS_SALU"]
                roof = dict(bound=binding.replace("_frac", ""), kernel="k_frame", achieved=n_inst / (us * B * 1e-6) / 1e9, peak=per_s / 1e9, unit="G wave-instructions/s", frac=fr[binding])
            roof.update(
                traffic=fr.get("hbm_bytes_per_launch"), hbm_frac=fr.get("hbm_frac"), valu_issue_frac=fr.get("valu_issue_frac"), salu_issue_frac=fr.get("salu_issue_frac"),
                packet_frac=fr.get("packet_frac"), contract_frac=fr.get("contract_frac"), l2_hit_rate=fr.get("l2_hit_rate"), valu_lane_utilisation=fr.get("valu_lane_utilisation"),
                machine_us_per_launch=us * B, kernel_ms=kernel_ms, launches_overlapping=kernel_ms * 1e3 / (us * B) if us else None, frames_timed=n_timed, frames_in_flight=F * B,
                algorithmic_bytes_per_launch=dict(packet=ab_launch.get("packet"), contract=ab_launch["contract"], packet_traversal=ab_launch.get("packet_traversal_bytes"),
                                                  shading=ab_launch.get("shading_bytes"), outputs=ab_launch.get("output_bytes")),
                pmc_source=entry["source"] if entry else None, pmc_kernel_source_sha16=entry.get("kernel_source_sha16") if entry else None,
                pmc_stale=(entry.get("kernel_source_sha16") != RL.kernel_source_hash()) if entry else None,
                note="frac = the binding roof among those MEASURED for this workload: wave-instructions issued per launch (committed rocprofv3 --pmc pass) over what the chip can issue in the launch's "
                     "share of machine time (ms_per_step: ~13 launches overlap, so kernel_ms, one launch's own span, is not that share).  hbm_frac = measured HBM traffic (2 x FETCH_SIZE + "
                     "WRITE_SIZE, Infinity-Cache hits included) / 8 TB/s.  packet_frac = the oracle's packet-level algorithmic bytes (a node / triangle once per 8x8-pixel packet, "
                     "canonical LBVH) / 8 TB/s.  contract_frac = SURVEY.md 8(d)'s per-ray algorithmic bytes / 8 TB/s: above 1 because the tree (~30 MB) is cache-resident and a packet "
                     "fetches a node once for 64 rays -- NOT an achieved bandwidth.  bound: null = no counter pass is committed for this workload (or N > 1): only the algorithmic fractions are given")

        line = {
            "metric": (("Mray/s (primary+shadow), Sponza-class 1080p" if (W, H) == (1920, 1080) else f"Mray/s (primary+shadow), Sponza-class {W}x{H}") if args.scene == "sponza"
                       else f"Mray/s (primary+shadow), Bistro-class {W}x{H}") if not glb else f"Mray/s (primary+shadow), {os.path.basename(glb)} {W}x{H}",
            "value": value, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "real glb" if glb else "synthetic",
            "config": {"workload": workload, "width": W, "height": H, "lights": len(lights),
                       "parallelism": ("single GPU" if world == 1 else f"screen tiles 32x32 over {G} tracing GPUs" + (" + 1 compositing GPU" if dedicated else (" (every rank assembles the frames f with f mod N = its rank)" if spread else f" (rank 0 composites too and traces {256 - relief}/256 of a share)"))
                                       + f", {transport}: gather of the {'B10G11R11 (4 B/px)' if packed else 'RGB32F HDR (12 B/px; alpha is the constant 1)'} colour tiles to {'the root of each frame' if spread else 'rank 0'}, {B} frames per launch, "
                                         f"{counts['launches_per_gather'] * B} frames per exchange") + f", {F * B} frames in flight"},
            "frames_per_s": args.steps / wall, "rays_per_frame": rays_total, "frames_in_flight": F * B, "frames_per_launch": B,
            "steady_state": dict(steady, value=rays_total / (steady["ms_per_step"] * 1e-3) / 1e6, unit="Mray/s") if steady else None,
            "camera_path": campath, "camera_jumps": camjumps, "moving_model": moving_model, "refit_ms": moving_model["refit_ms"] if moving_model else None,
            "stage_ms": stage, "frame_ms_one_frame_alone": alone, "build_ms": st["build_ms"], "settle_frames": settle_launches * B,
            "tuning": tuning, "gathered_frame_equals_single_gpu_frame": frame_ok, "gathers": counts["gathers"] if counts else None,
            **({("value_rank0_root" if other["placement"] == "rank0" else "value_spread_roots"): other.get("value"), "other_placement": other} if other else {}),
            "roofline": roof, "cpu_baseline": cpu,
            "degraded": bool(bailing), "value_placement": (main_pl if world > 1 else None),   # degraded: the second placement was given up (its stacks are on stderr) -- `value` is then value_placement's, whatever --roots asked for
        }
        print(json.dumps(line), flush=True)
        if world > 1 and not bailing:
            dist.destroy_process_group()

    # N > 1: both placements of the assembled frames back to back, each a whole job of its own (context, communicator, settle, warm-up, K timed steps, frame check).
    # The one with the single ncclGather runs first; the second one runs under a timer: should it not finish -- no exchange with more than one rank has ever run on
    # a fabric -- rank 0 still prints the line of the first and every rank leaves with exit code 0, instead of the run ending with nothing at the watchdog.
    order = placements if world == 1 else sorted(placements, key=lambda p_: p_ != "rank0")
    jobs = {}

    # Whether the second placement is given up is AGREED between the ranks, through a small key-value store of the job's own (rank 0 serves it): a rank that finishes its second
    # placement says so and waits until every rank has said so -- or until one has bailed, then it bails too -- before it enters any later collective; a rank whose timer fires
    # says "bailed" first.  (Round 3: every rank decided by its own timer; one that finished a moment before its peers' timers fired went on into a collective they had left.)
    store = None
    if world > 1 and len(order) > 1 and args.second_placement_seconds > 0:
        from datetime import timedelta
        store = dist.TCPStore("127.0.0.1", int(os.environ.get("MASTER_PORT", "29500")) + 17, world, rank == 0, timeout=timedelta(seconds=30))

    def bail(pl_):   # (a timer thread, or the main thread of a rank that learns of a peer's bail)
        try:
            if store is not None:
                store.add("bailed", 1)
        except Exception:
            pass
        import faulthandler
        print(f"[bench] rank {rank}: placement '{pl_}' did not finish within {args.second_placement_seconds:.0f} s; this rank's threads:", file=sys.stderr, flush=True)
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)   # where every rank stood: the cause is read from these stacks, never from a retry
        if rank == 0:
            finish(order[0], dict(placement=pl_, error=f"did not finish within {args.second_placement_seconds:.0f} s: this line is the other placement's alone"), bailing=True)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)   # rank 0 has printed a valid line that says "degraded": true and whose placement `value` is; a non-zero code on any rank would make the launcher fail the
                      # whole job and the driver drop that measurement with it (the stacks above are the diagnosis)

    def agree_done(pl_):   # every rank finished the second placement, or all give it up together
        if store is None:
            return
        try:
            store.add("done", 1)
            while True:
                if int(store.add("bailed", 0)) > 0:
                    bail(pl_)
                if int(store.add("done", 0)) >= world:
                    return
                time.sleep(0.05)
        except SystemExit:
            raise
        except Exception:      # the store went away with rank 0: it bailed
            bail(pl_)
    for i_, pl in enumerate(order):
        guard = None
        if i_ > 0 and args.second_placement_seconds > 0:
            import threading
            guard = threading.Timer(args.second_placement_seconds, bail, args=(pl,))
            guard.daemon = True
            guard.start()
        jobs[pl] = run_job(pl)
        if guard:
            guard.cancel()
            agree_done(pl)

    def describe(o_, pl_):
        return dict(placement=pl_, value=o_["value"], unit="Mray/s", ms_per_step=o_["ms_per_step"], gathered_frame_equals_single_gpu_frame=o_["frame_ok"], gathers=o_["counts"]["gathers"],
                    frames_per_exchange=o_["counts"]["launches_per_gather"] * o_["B"], rank0_share_of_an_equal_share=f"{256 - o_['relief']}/256")
    main_pl = placements[0]           # `value` is the placement --roots names
    finish(main_pl, describe(jobs[placements[1]], placements[1]) if len(placements) > 1 else None)


def workload_name(sc, glb, W, H, lights, shadow_total, args):
    scene = (f"{os.path.basename(glb)} through art_scene_add_glb (" if glb else ("bistro_like(seed=0xB157, " if args.scene == "bistro" else "sponza_like(seed=0x5A0A, "))
    return f"{scene}{sc.n_tris} triangles, {len(sc.primitives)} primitives) {W}x{H}, {len(lights)} light(s), {int(shadow_total)} shadow rays/frame" + (f", {args.ao} AO rays per hit pixel" if args.ao else "")


if __name__ == "__main__":
    main()
