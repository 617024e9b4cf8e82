"""CPU tests of the host side: the C ABI library loads and exports every symbol include/art.h declares, fails loudly
without a device, scene generators, host maths behind the ABI (no device needed), screen-tile sharding incl. a
world_size-2 gloo run."""
import ctypes as C
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from araytracingjourney_amd import _lib
    L = _lib.load()
    for header, table in (("art.h", _lib.SYMBOLS), ("art_parity.h", _lib.PARITY_SYMBOLS)):   # the boundary; the parity / rehearsal / measurement surface
        hdr = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
        declared = set(re.findall(r"\b(art_[a-z_0-9]+)\s*\(", hdr))
        assert declared == set(table), (header, declared ^ set(table))
        for name in declared:
            assert getattr(L, name) is not None
    assert not set(_lib.SYMBOLS) & set(_lib.PARITY_SYMBOLS)


def test_struct_layouts_match_the_reference_contracts():
    from araytracingjourney_amd import _lib
    assert C.sizeof(_lib.ArtVertex) == 48      # gltf_model_reader.rs:176-199
    assert C.sizeof(_lib.ArtLight) == 80       # lights.rs:69-82
    assert C.sizeof(_lib.ArtCamera) == 268     # vk_camera.rs:9-16
    assert _lib.ArtLight.type.offset == 12 and _lib.ArtLight.casts_shadows.offset == 28 and _lib.ArtLight.umbra_angle.offset == 76


def test_no_device_fails_loudly_not_silently():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from araytracingjourney_amd import _lib, renderer
    with pytest.raises(_lib.ArtError) as e:
        renderer.Renderer((64, 64))
    assert e.value.code == _lib.ART_E_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_product_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "araytracingjourney_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in txt.lower() or f == "scenes.py" and False, f"{f} mentions the oracle"


def test_host_maths_behind_the_abi_matches_the_oracle(orc):
    """art_camera_from_params / art_light_* run on the host: comparable here without a GPU"""
    from araytracingjourney_amd import renderer
    cam = renderer.Camera((-1.2, 0.35, 0.0), (1.0, -0.05, 0.1), 16 / 9, math.pi / 2, 0.1, 1000.0)
    a = cam.update_host_buffer()
    b = orc.camera_from_params((-1.2, 0.35, 0.0), (1.0, -0.05, 0.1), 16 / 9, math.pi / 2, 0.1, 1000.0)
    assert bytes(a) == bytes(b)
    ls = renderer.Lights()
    from araytracingjourney_amd import scenes
    for d in scenes.sponza_lights(4):
        ls.push_dict(d)
    arr, n = ls.copy_lights_shader_data()
    ref = orc.make_lights(scenes.sponza_lights(4))   # serialisation order point, spot, directional, area (lights.rs:24-47)
    assert n == 4 and [arr[i].type for i in range(4)] == [0, 1, 2, 3]
    for i in range(4):
        assert bytes(arr[i]) == bytes(ref[i])


def test_cpp_host_mirror_builds_and_passes_its_host_checks():
    """araytracingjourney_amd/host/art_renderer.hpp: the C++ twin of renderer.rs / lights.rs / vk_camera.rs over the C ABI"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    r = subprocess.run([os.path.join(ROOT, "examples", "host_mirror_demo"), "check"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "HOST_MIRROR_OK" in r.stdout, r.stdout + r.stderr


def test_scene_generators(get_scene):
    c = get_scene("cornell")
    assert c.n_tris == 34 and len(c.primitives) == 3 and all(p.indices.dtype == np.uint16 for p in c.primitives)
    assert abs(max(np.linalg.norm(p.verts[:, :3], axis=1).max() for p in c.primitives) - 1.0) < 1e-6      # unit-ball normalisation
    s = get_scene("sponza_like", 1.0)
    assert abs(s.n_tris - 262144) / 262144 < 0.01 and len(s.primitives) == 25
    assert sum(p.indices.dtype == np.uint32 for p in s.primitives) == 1 and max(p.verts.shape[0] for p in s.primitives) > 65535
    for p in s.primitives:
        v = p.verts
        assert np.isfinite(v).all() and p.tex.shape == (3, 256, 256, 4) and int(p.indices.max()) < v.shape[0]
        assert np.allclose(np.linalg.norm(v[:, 5:8], axis=1), 1, atol=1e-5) and np.allclose(np.linalg.norm(v[:, 8:11], axis=1), 1, atol=1e-5)
        assert np.abs((v[:, 5:8] * v[:, 8:11]).sum(1)).max() < 1e-5 and set(np.unique(v[:, 11])) <= {-1.0, 1.0}
        assert v[:, 3:5].min() >= 0 and v[:, 3:5].max() <= 4.0
        assert p.tex[1, ..., 1].min() >= 51                                                                   # roughness >= 0.2: no BRDF NaN hazards
    import hashlib
    assert hashlib.sha256(get_scene("sponza_like", 0.05).primitives[0].verts.tobytes()).hexdigest() == \
        hashlib.sha256(__import__("araytracingjourney_amd.scenes", fromlist=["x"]).sponza_like(0.05).primitives[0].verts.tobytes()).hexdigest()


def test_shard_layout_partitions_the_frame():
    from araytracingjourney_amd import sharding
    for (w, h, g) in [(1920, 1080, 8), (1920, 1080, 2), (3840, 2160, 4), (200, 136, 3), (33, 31, 8), (64, 64, 1)]:
        all_tiles, pads = [], set()
        for r in range(g):
            t, padded = sharding.shard_layout(w, h, g, r)
            assert len(t) <= padded and np.all(np.diff(t.astype(np.int64)) > 0)
            all_tiles += t.tolist()
            pads.add(padded)
        n = ((w + 31) // 32) * ((h + 31) // 32)
        assert sorted(all_tiles) == list(range(n)) and len(pads) == 1
        frame = np.arange(w * h * 4, dtype=np.float32).reshape(h, w, 4)
        gathered = np.stack([sharding.tile_host(frame, g, r) for r in range(g)])
        assert np.array_equal(sharding.untile_host(gathered, w, h, g), frame)
    t, padded = sharding.shard_layout(1920, 1080, 8, 0)
    assert padded - len(t) <= 1 and padded * 8 - 2040 <= 8          # near-equal shares: load balance


def test_root_relief_moves_tiles_from_shard_0_to_the_others():
    """ArtConfig.root_relief (art_shard_layout's root_relief): the compositing rank's share shrinks by per_256 / 256, the others grow evenly, the frame stays a
    partition; the value is an argument -- nothing process-wide is left behind"""
    from araytracingjourney_amd import sharding, _lib
    w, h, g = 1920, 1080, 8
    counts, all_tiles = [], []
    for r in range(g):
        t, padded = sharding.shard_layout(w, h, g, r, root_relief=64)
        counts.append(len(t)); all_tiles += t.tolist()
    assert sorted(all_tiles) == list(range(2040))
    assert 170 <= counts[0] <= 210 and max(counts[1:]) - min(counts[1:]) <= 2 and padded == max(counts)    # 255 * 3/4 = 191 +- the subset's luck
    frame = np.arange(w * h * 4, dtype=np.float32).reshape(h, w, 4)
    gathered = np.stack([sharding.tile_host(frame, g, r, 64) for r in range(g)])
    assert np.array_equal(sharding.untile_host(gathered, w, h, g, 64), frame)
    with pytest.raises(_lib.ArtError):
        sharding.shard_layout(w, h, g, 0, root_relief=256)
    assert len(sharding.shard_layout(w, h, g, 0)[0]) == 255


_GLOO_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from araytracingjourney_amd import sharding
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
w, h = 200, 136
frame = (np.arange(w * h * 4, dtype=np.float32).reshape(h, w, 4) * 0.5)          # what every rank would render
mine = torch.from_numpy(sharding.tile_host(frame, world, rank))                   # this rank's compact tile buffer
owned, padded = sharding.shard_layout(w, h, world, rank)
assert mine.shape[0] == padded
gathered = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
dist.gather(mine, gathered, dst=0)                                                 # the one exchange step of a frame
rays = torch.tensor([float(len(owned) * 1024)], dtype=torch.float64)
dist.all_reduce(rays)                                                              # whole-job ray count, as bench.py does
packed = (np.arange(w * h, dtype=np.uint32).reshape(h, w) * np.uint32(2654435761))   # the B10G11R11 payload: one word per pixel
pmine = torch.from_numpy(sharding.tile_host(packed, world, rank).view(np.int32))
pgathered = [torch.empty_like(pmine) for _ in range(world)] if rank == 0 else None
dist.gather(pmine, pgathered, dst=0)
if rank == 0:
    got = sharding.untile_host(torch.stack(gathered).numpy(), w, h, world)
    assert np.array_equal(got, frame)
    assert np.array_equal(sharding.untile_host(torch.stack(pgathered).numpy().view(np.uint32), w, h, world), packed)
    assert rays.item() == ((w + 31) // 32) * ((h + 31) // 32) * 1024
    print("GLOO_OK")
# dedicated compositor (bench.py from 4 GPUs on): rank 0 traces nothing, ranks 1.. are the shards 0.. of the frame, several frames per gather
G = world - 1
if G >= 1:
    F = 3
    frames = [frame * np.float32(1 + j) for j in range(F)]
    if rank == 0:
        _, padded = sharding.shard_layout(w, h, G, 0)
        mine = torch.zeros((F, padded, 32, 32, 4), dtype=torch.float32)              # a dummy of the right size
    else:
        mine = torch.from_numpy(np.stack([sharding.tile_host(f, G, rank - 1) for f in frames]))
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, parts, dst=0)
    if rank == 0:
        got = torch.stack(parts[1:]).numpy()                                          # [shard][frame][padded]...
        for j in range(F):
            assert np.array_equal(sharding.untile_host(got[:, j], w, h, G), frames[j]), j
        print("COMPOSITOR_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gather_over_gloo(tmp_path):
    """the N>1 plumbing of bench.py (shard -> gather to rank 0 -> un-tile) with world_size 2 on CPUs"""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                        str(script), ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "GLOO_OK" in r.stdout and "COMPOSITOR_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_three_rank_dedicated_compositor_over_gloo(tmp_path):
    """world_size 3: rank 0 composites, ranks 1-2 are the two shards (the N >= 4 mode of bench.py), frames batched per gather"""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1", "--master-port", "29542",
                        str(script), ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "GLOO_OK" in r.stdout and "COMPOSITOR_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_and_tools_parse_and_bench_refuses_to_run_without_a_gpu():
    """bench.py: its argument parser loads, and without a GPU it stops with a message instead of falling back to anything; tools compile"""
    import glob, py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--frames-per-launch" in out.stdout and "--root-relief" in out.stdout
    for f in glob.glob(os.path.join(root, "tools", "*.py")) + [os.path.join(root, "__graft_entry__.py")]:
        py_compile.compile(f, doraise=True)
    import torch
    if not torch.cuda.is_available():
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1"], capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "no CPU fallback" in (out.stderr + out.stdout)


def test_model_residency_state_machine():
    """vk_model.rs:334-345 with the camera positions of the reference's own test (vk_model.rs:1082-1152): a model goes
    Storage / Host / Device by the distance between the camera and its bounding sphere (<= 10: Device, <= 20: Host)"""
    from araytracingjourney_amd import renderer as R
    m = R.Model([0], R.Sphere((0.0, 0.0, 0.0), 1.0))
    assert m.state == R.HOST                                   # VkModel::new: Storage -> Host
    m.update_model_status((100.0, 100.0, 100.0))
    assert m.state == R.STORAGE and not m.needs_command_buffer_submission()
    m.update_model_status((7.0, 7.0, 7.0))                      # |(7,7,7)| - 1 = 11.1
    assert m.state == R.HOST and not m.needs_command_buffer_submission()
    m.update_model_status((3.0, 3.0, 3.0))                      # 4.2
    assert m.state == R.DEVICE and m.needs_command_buffer_submission()
    m.reset_command_buffer_submission_status()
    m.update_model_status((7.0, 7.0, 7.0))
    assert m.state == R.HOST and m.needs_command_buffer_submission()   # leaving Device needs a submission too (vk_model.rs:1152-1154)
    s = R.Sphere((1.0, 0.0, 0.0), 2.0).transform([[2, 0, 0, 5], [0, 3, 0, 0], [0, 0, 1, 0]])   # model_reader.rs:128-141
    assert np.allclose(s.center, (7.0, 0.0, 0.0)) and s.radius == 6.0
    assert abs(R.Sphere((0, 0, 0), 1.0).get_distance_from_point((0, 3, 4)) - 4.0) < 1e-6


def test_rust_binding_is_generated_from_the_header_and_complete():
    """bindings/art_sys.rs (the extern "C" block a maintainer of the reference would add) is what tools/gen_rust_bindings.py makes of
    include/art.h today, and names every function the header declares and the ctypes table binds; no Rust toolchain here, so: generated, not compiled"""
    import re, subprocess, sys
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_bindings.py"), "--check"], capture_output=True, text=True).returncode == 0
    hdr = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "art.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(art_\w+)\s*\(", hdr))
    rs = open(os.path.join(ROOT, "bindings", "art_sys.rs")).read()
    bound = set(re.findall(r"pub fn (art_\w+)\(", rs))
    from araytracingjourney_amd import _lib
    assert declared == bound == set(_lib.SYMBOLS), (declared ^ bound, declared ^ set(_lib.SYMBOLS))
    assert not bound & set(_lib.PARITY_SYMBOLS) and "ArtTuning" not in rs          # the parity surface (include/art_parity.h) is not what a maintainer binds
    for name, size in (("ArtVertex", 48), ("ArtLight", 80), ("ArtCamera", 268), ("ArtConfig", 36)):
        assert f"size_of::<{name}>() == {size})" in rs
    assert "#[repr(C, packed)] #[derive(Clone, Copy)]\npub struct ArtCamera" in rs


def test_bench_started_plainly_spawns_its_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE (how the driver starts its one-GPU run): the parent starts torch.distributed.run with two fresh children
    as a CHILD process and relays its exit code -- here, without a GPU, both ranks get as far as "needs an MI355X" (libart has no CPU fallback), which is the proof that both ran"""
    import subprocess, sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU box runs this as tests/test_mgpu.py::test_many_ranks_with_the_drivers_arguments_through_both_placements")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0 and "starting the ranks as a child process" in out.stderr
    assert out.stderr.count("bench.py needs an MI355X") >= 2, out.stderr[-3000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_roofline_is_rederived_from_the_committed_counter_files(tmp_path):
    """tools/roofline.py --tag round4 on the files under profiles/ (rocprofv3 counter passes of configs 2-5, the bench lines they belong to): every fraction the docs quote
    comes out of it again -- the frame kernel bound by vector-instruction issue, the AO launch by the texture addresser (tools/pmc_ta.sh's counters, round 4b)"""
    import json
    import shutil
    work = tmp_path / "repo"
    for d in ("profiles", "tools", os.path.join("tests", "golden")):
        shutil.copytree(os.path.join(ROOT, d), work / d, ignore=shutil.ignore_patterns("__pycache__", "*.glb", "*.npz", "*.npy"))
    shutil.copytree(os.path.join(ROOT, "araytracingjourney_amd", "csrc"), work / "araytracingjourney_amd" / "csrc", ignore=shutil.ignore_patterns("*.o", "*.so"))
    out = subprocess.run([sys.executable, str(work / "tools" / "roofline.py"), "--tag", "round4"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout)
    assert set(r) == {"c2", "c3", "c4", "c5"}
    for k in ("c2", "c3", "c4"):
        assert r[k]["binding_roof"] == "valu_issue" and 0.6 < r[k]["valu_issue_frac"] < 0.8 and r[k]["hbm_frac"] < 0.25
    assert 0.05 < r["c2"]["ta_busy_frac"] < 0.12                      # the packets' nodes come through the scalar cache
    assert r["c5"]["binding_roof"] == "ta_busy" and 0.8 < r["c5"]["ta_busy_frac"] < 0.92 and 20 < r["c5"]["ta_cycles_per_load_instruction"] < 30
    committed = json.load(open(os.path.join(ROOT, "profiles", "round4_roofline.json")))
    for k in r:
        assert abs(committed[k]["valu_issue_frac"] - r[k]["valu_issue_frac"]) < 1e-9, k      # the committed summary is this output
