"""The sharded frame behind the C ABI (include/art.h, art_mgpu_*): host-only pieces on the CPU, the frame itself on the GPU --
one rank over RCCL (ART_FLAG_TILE_OUTPUT), and 2 / 3 rank jobs of fresh child processes with the collective replaced by a host function
over gloo (RCCL refuses two ranks on one device; the 8-GPU run is the driver's)."""
import ctypes as C
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mgpu_shard_rule_and_argument_errors():
    from araytracingjourney_amd import _lib, renderer
    assert [renderer.mgpu_shard(r, 4) for r in range(4)] == [(0, 4), (1, 4), (2, 4), (3, 4)]
    assert [renderer.mgpu_shard(r, 4, dedicated=True) for r in range(4)] == [(0, 3), (0, 3), (1, 3), (2, 3)]   # rank 0 composites, ranks 1.. are shards 0..
    assert renderer.mgpu_shard(0, 1) == (0, 1)
    L = _lib.load()
    sr, sc = C.c_uint32(), C.c_uint32()
    assert L.art_mgpu_shard(4, 4, 0, C.byref(sr), C.byref(sc)) == _lib.ART_E_INVALID          # rank >= world
    assert L.art_mgpu_shard(0, 1, 1, C.byref(sr), C.byref(sc)) == _lib.ART_E_INVALID          # a dedicated compositor alone
    assert L.art_mgpu_shard(0, 0, 0, C.byref(sr), C.byref(sc)) == _lib.ART_E_INVALID
    h = C.c_void_p()
    assert L.art_mgpu_create(None, None, None, C.byref(h)) == _lib.ART_E_INVALID and b"null" in L.art_last_error()
    for fn in (L.art_mgpu_trace, L.art_mgpu_flush):
        assert fn(None) == _lib.ART_E_INVALID
    assert L.art_mgpu_destroy(None) == _lib.ART_OK
    assert C.sizeof(_lib.ArtMgpuConfig) == 48 and C.sizeof(_lib.ArtLayout) == 40


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("spread", [False, True], ids=["rank0", "spread"])
@pytest.mark.parametrize("packed", [False, True])
def test_one_rank_job_over_rccl_assembles_the_frame(get_scene, packed, spread):
    """RCCL inside libart on the one GPU there is: ncclGetUniqueId, ncclCommInitRank (1 rank), the exchange of every group -- ncclGather to rank 0, or with
    spread roots ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd (to itself here: the entry points and their argument order are exercised, the fabric is
    not) --, un-tile; frames with a moving camera through a ring of 4 slots x 2 frames per launch, 2 launches per gather; every flushed frame equals the
    unsharded render bit for bit (RGBA32F: the HDR buffer; packed: the B10G11R11 colour image)"""
    from araytracingjourney_amd import renderer as R, _lib
    sc = get_scene("sponza_like", 0.12)
    w, h, F, B = 480, 270, 4, 2
    whole = R.renderer_for_scene(sc, (w, h))
    r = R.renderer_for_scene(sc, (w, h), frames_in_flight=F, tile_output=True, packed_tiles=packed)
    lay = r.layout()
    assert (lay["shard_count"], lay["tiles_owned"], lay["tiles_padded"], lay["tile_bytes"]) == (1, 15 * 9, 15 * 9, 4096 if packed else 12288)
    r.set_frames_per_launch(B)
    r.upload_state()
    for _ in range(3):
        r.trace()                                       # frames traced before the job starts: the ring is rewound at create
    mg = R.MultiGpu(r, 0, 1, unique_id=R.mgpu_unique_id(), launches_per_gather=2, spread=spread)
    p0 = sc.camera["pos"]
    for i in range(11):                                 # 11 launches: groups of 2, a trip's wrap, a partial group at the flush
        cams = [R.Camera((p0[0] + 0.01 * (2 * i + b), p0[1], p0[2] + 0.004 * i), r.camera_mut().dir(), w / h, r.camera_mut().fovy(), 0.1, 1000.0) for b in range(B)]
        r.set_camera_batch(cams)
        mg.trace()
        if i == 0:                                      # the launch waits in an open group (or its group for its exchange): a control plane must not be entered now
            assert mg.pending() != (0, 0)
            with pytest.raises(RuntimeError, match="flush"):
                mg.assert_quiescent()
        if i in (0, 4, 10):
            mg.flush()
            assert mg.pending() == (0, 0)
            mg.assert_quiescent()
            got = mg.read_frame()
            whole._camera = cams[-1]
            whole.render_frame()
            if packed:
                whole.present()
                assert np.array_equal(got, whole.read_packed()[0]), i
            else:
                assert np.array_equal(got.view(np.uint32), whole.read_color().view(np.uint32)), i
    r.timestamp_mark(0)                                  # art_timestamp_*: device time between two points of the frame streams
    r.trace(); r.trace()
    r.timestamp_mark(1)
    assert 0.0 < r.timestamp_elapsed_ms() < 100.0
    c = mg.counts()
    assert c["launches_traced"] == 11 and c["launches_per_gather"] == 2 and c["gathers"] >= 6
    with pytest.raises(_lib.ArtError):                  # a context that writes no tiles cannot take part
        R.MultiGpu(whole, 0, 1, unique_id=R.mgpu_unique_id())
    with pytest.raises(_lib.ArtError):                  # nor one whose shard is not the rank's
        R.MultiGpu(r, 1, 2, unique_id=R.mgpu_unique_id())
    mg.close(); r.close(); whole.close()


@pytest.mark.gpu
def test_a_moving_model_through_the_sharded_frame(get_scene):
    """art_scene_set_model_matrix under art_mgpu_*: every rank makes the same calls, each refits its copy of the scene in front of its next launch; the frames
    the job assembles (one rank over RCCL here, spread roots) are the unsharded renders of the moved scene, bit for bit, also with launches in flight"""
    from araytracingjourney_amd import renderer as R
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_parity import _pose
    sc = get_scene("sponza_like", 0.12)
    w, h, F = 480, 270, 4

    def make(**kw):
        r = R.Renderer((w, h), **kw)
        r.add_model(sc.primitives[:-1]); r.add_model(sc.primitives[-1:])
        cam = r.camera_mut()
        cam.set_pos(sc.camera["pos"]); cam.set_dir(sc.camera["dir"]); cam.set_fovy(sc.camera["fovy"]); cam.set_znear(sc.camera["znear"]); cam.set_zfar(sc.camera["zfar"])
        for d in sc.lights:
            r.lights_mut().push_dict(d)
        r.prepare_first_frame(); r.upload_state()
        return r
    whole, r = make(), make(frames_in_flight=F, tile_output=True)
    mg = R.MultiGpu(r, 0, 1, unique_id=R.mgpu_unique_id(), launches_per_gather=2, spread=True)
    for i in range(1, 10):
        m = _pose(sc.primitives[-1].model, i)
        r.models_mut()[1].set_model_matrix(m)
        mg.trace()
        if i in (1, 6, 9):
            mg.flush()
            whole.models_mut()[1].set_model_matrix(m)
            whole.render_frame()
            assert np.array_equal(mg.read_frame().view(np.uint32), whole.read_color().view(np.uint32)), i
    assert r.stats()["refits"] == 9
    mg.close(); r.close(); whole.close()


@pytest.mark.gpu
def test_a_failed_exchange_is_reported_and_latched(get_scene):
    """a transport that fails (the host hook here; an RCCL error alike) loses its group's frames and leaves the ranks out of step: the call that submitted the
    exchange returns the error, and so does every later trace and flush -- none reports success for frames that never arrived, none waits for the lost group"""
    from araytracingjourney_amd import renderer as R, _lib
    sc = get_scene("cornell")
    r = R.renderer_for_scene(sc, (96, 64), frames_in_flight=4, tile_output=True)
    r.upload_state()
    calls = []

    def link_down(send, nbytes, recv, root, stream):
        calls.append(nbytes)
        raise RuntimeError("link down")
    mg = R.MultiGpu(r, 0, 1, launches_per_gather=2, exchange=link_down)
    with pytest.raises(_lib.ArtError) as e:
        for _ in range(12):
            mg.trace()
        mg.flush()
    assert "exchange function failed" in str(e.value) and len(calls) == 1
    for fn in (mg.trace, mg.flush, mg.trace):
        with pytest.raises(_lib.ArtError) as e:
            fn()
        assert "earlier exchange failed" in str(e.value)
    assert len(calls) == 1
    mg.close(); r.close()


def _run_job(cmd, env, seconds):
    """the job in a process group of its own: a job that hangs fails its test and is killed with all its ranks, it does not take the run with it"""
    import signal
    import types
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
    try:
        so, se = p.communicate(timeout=seconds)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        so, se = p.communicate()
        return types.SimpleNamespace(returncode=-9, stdout=so, stderr=se + f"\n[killed after {seconds} s]")
    return types.SimpleNamespace(returncode=p.returncode, stdout=so, stderr=se)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,extra", [(2, []), (3, ["--gather", "packed"]), (3, ["--gather-launches", "3", "--frames-per-launch", "2"]), (2, ["--ao", "4"]),
                                         (2, ["--roots", "rank0"]), (2, ["--roots", "rank0", "--gather", "packed", "--one-placement"]), (3, ["--compositor", "dedicated"]), (3, ["--roots", "rank0", "--gather-launches", "3", "--frames-per-launch", "2"])],
                         ids=["spread", "spread-packed-3", "spread-groups-of-3x2", "spread-ao", "rank0", "rank0-packed-only", "dedicated", "rank0-groups-of-3x2"])
def test_sharded_bench_job_of_child_processes_gathers_the_single_gpu_frame(ranks, extra):
    """bench.py as the driver launches it (torch.distributed.run, one process per rank), the ranks sharing the one GPU and the collective
    going through gloo: the C++ loop of art_mgpu_* -- tile-buffer rings, host-gated groups, un-tile -- with real concurrency, for both placements
    of the assembled frames (spread over the ranks: bench.py's default; all on rank 0); every rank that assembles frames checks the newest one
    it holds against an unsharded render, bit for bit.  One invocation times BOTH placements back to back -- `value` with the one --roots names, the
    other's rate beside it -- unless --one-placement (or a dedicated compositor, which implies rank 0) says otherwise"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--backend", "gloo", "--steps", "24", "--warmup", "4", "--no-cpu-baseline", "--detail", "0.12",
           "--width", "640", "--height", "360", "--frames-in-flight", "6", "--settle-seconds", "0.05", "--watchdog-seconds", "200"] + extra
    out = _run_job(cmd, env, 300)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads(lines[0])
    assert line["gathered_frame_equals_single_gpu_frame"] is True
    assert line["n_gpus"] == ranks and line["gathers"] >= 4 and line["value"] > 0
    if "--one-placement" in extra or "dedicated" in extra:
        assert "other_placement" not in line
    else:
        first = "rank0" if "rank0" in extra else "spread"
        other = line["other_placement"]
        assert other["placement"] == ("spread" if first == "rank0" else "rank0") and other["gathered_frame_equals_single_gpu_frame"] is True and other["gathers"] >= 4
        assert line["value_spread_roots" if first == "rank0" else "value_rank0_root"] == other["value"] > 0
        assert ("rank 0 composites too" in line["config"]["parallelism"]) == (first == "rank0")
    assert ("B10G11R11" in line["config"]["parallelism"]) == ("packed" in extra) and ("RGB32F HDR" in line["config"]["parallelism"]) == ("packed" not in extra)


@pytest.mark.gpu
def test_a_second_placement_that_hangs_still_leaves_the_first_ones_line():
    """no exchange with more than one rank has ever run on a fabric: should the second placement of a `bench.py --gpus N` run never finish, rank 0 prints the
    first placement's line (the single ncclGather's: it runs first) with the failure noted, and every rank exits 0 -- rehearsed with an exchange that sleeps"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "24", "--warmup", "4", "--no-cpu-baseline", "--detail", "0.12",
           "--width", "640", "--height", "360", "--frames-in-flight", "6", "--settle-seconds", "0.05", "--watchdog-seconds", "200", "--rehearse-hang", "--second-placement-seconds", "15"]
    out = _run_job(cmd, env, 240)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads(lines[0])
    assert line["value"] > 0 and line["gathered_frame_equals_single_gpu_frame"] is True and "rank 0 composites too" in line["config"]["parallelism"]
    assert line["other_placement"]["placement"] == "spread" and "did not finish" in line["other_placement"]["error"] and line["value_spread_roots"] is None
    assert line["degraded"] is True and line["value_placement"] == "rank0"          # the line says whose `value` it carries, whatever --roots asked for
    assert out.stderr.count("did not finish within 15 s; this rank's threads") == 2 and "gloo_gather" in out.stderr   # every rank left its stacks: where it hung


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--frames-per-launch", "2"]], ids=["4-frames-per-launch", "2-frames-per-launch-roots-alternate"])
def test_many_ranks_with_the_drivers_arguments_through_both_placements(extra):
    """the driver's own `--gpus N --steps 20 --warmup 5`, started PLAINLY (bench.py spawns its ranks): one launch per exchange, so a rank roots at most ONE frame of a group
    (nf_cap = 1, art_mgpu.hip:112-128); with 2 frames per launch over 4 ranks a group's frames fall to ranks 0-1 and 2-3 in turn -- what 4 frames per launch do to ranks 0-3 and
    4-7 of the driver's eight (the corner no smaller job reaches).  Four child processes share the one GPU: six processes at most may hold it at once on the test box, and this
    test process and the launcher are two of them (ART_TEST_RANKS overrides: 8 on a box without that guard).  The exchange goes through the gloo hook, BOTH placements are timed,
    every assembling rank checks the newest frame it holds against an unsharded render bit for bit"""
    ranks = int(os.environ.get("ART_TEST_RANKS", "4"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="4")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "20", "--warmup", "5", "--backend", "gloo", "--no-cpu-baseline", "--detail", "0.12",
           "--width", "640", "--height", "360", "--settle-seconds", "0.05", "--watchdog-seconds", "250"] + extra
    out = _run_job(cmd, env, 420)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-6000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == ranks and line["steps"] == 20 and line["warmup"] == 5 and line["frames_per_launch"] == (2 if extra else 4) and line["degraded"] is False and line["value_placement"] == "spread"
    assert line["gathered_frame_equals_single_gpu_frame"] is True and line["other_placement"]["gathered_frame_equals_single_gpu_frame"] is True
    assert line["gathers"] >= 5 and line["other_placement"]["gathers"] >= 5 and line["value"] > 0 and line["value_rank0_root"] > 0
    assert "starting the ranks as a child process" in out.stderr


@pytest.mark.gpu
def test_config4_eight_way_split_assembles_the_full_size_frame(get_scene):
    """BASELINE config 4's split at its real size on the one GPU there is: eight shard contexts of the 2.8 M-triangle scene at 1920x1080, each
    tracing its tiles; their compact tile buffers laid out as ncclGather leaves them on rank 0 and un-tiled by one launch give the unsharded
    frame bit for bit; every shard owns 255 or 256 of the 2040 tiles"""
    import torch
    from araytracingjourney_amd import renderer as R
    sc = get_scene("bistro_like", 1.0)
    w, h, G = 1920, 1080, 8
    whole = R.renderer_for_scene(sc, (w, h))
    whole.render_frame()
    want = whole.read_color()
    st_whole = whole.stats()
    whole.close()
    bufs, owned, rays = [], [], 0
    for k in range(G):                                   # one at a time: eight copies of the scene need not be resident together
        s = R.renderer_for_scene(sc, (w, h), shard=(k, G))
        s.render_frame()
        bufs.append(torch.from_numpy(s.read_color_tiles()).cuda())
        owned.append(s.shard_tile_count()[0])
        rays += s.stats()["shadow_rays"]
        if k < G - 1:
            s.close()
    gathered = torch.stack(bufs).contiguous()            # [rank][padded tiles][32][32][4]
    torch.cuda.synchronize()
    s.untile_gathered(gathered.data_ptr(), G)            # the last shard's context un-tiles (any shard's layout tables are the job's)
    got = s.read_color()
    s.close()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert sum(owned) == 60 * 34 and max(owned) - min(owned) <= 1 and rays == st_whole["shadow_rays"]


@pytest.mark.gpu
def test_bench_takes_a_glb_through_the_real_ingest(tmp_path, get_scene):
    """bench.py --glb: the scene comes through art_glb_open + art_scene_add_glb (renderer.rs:346), the line says data: "real glb", the GPU's ray
    counts equal the oracle's on the reader's own output (asserted inside bench.py), and the roofline's packet-level bytes are counted for it"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from glb_writer import write_glb
    sc = get_scene("sponza_like", 0.05)
    path = tmp_path / "atrium.glb"
    write_glb(str(path), sc.primitives, png_modes=("RGBA", "RGBA", "RGBA"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--glb", str(path), "--steps", "16", "--warmup", "4", "--width", "640", "--height", "360",
                          "--cpu-seconds", "0.5", "--camera-path", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads(lines[0])
    assert line["data"] == "real glb" and "atrium.glb through art_scene_add_glb" in line["config"]["workload"] and line["value"] > 0
    assert line["cpu_baseline"]["kind"] == "port" and line["roofline"]["bound"] is None and 0 < line["roofline"]["packet_frac"] < 1   # no counter pass for this workload: no roof is named
