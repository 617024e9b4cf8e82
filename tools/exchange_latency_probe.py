"""how long after the last frame of a gather group has finished does the group's exchange (a device copy standing in for the RCCL gather)
finish, on a GPU that keeps tracing?  usage: exchange_latency_probe.py G GB   (env PF frames in flight)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "22")
sys.path.insert(0, ".")
import torch
from araytracingjourney_amd import renderer, scenes
G, GB, F = int(sys.argv[1]), int(sys.argv[2]), int(os.environ.get("PF", "20"))
sc = scenes.sponza_like()
r = renderer.renderer_for_scene(sc, (1920, 1080), shard=(0, G), frames_in_flight=F, packed_tiles=True)
r.upload_state()
xs, ts = torch.cuda.Stream(), torch.cuda.Stream()
owned, padded = r.shard_tile_count()
tiles = torch.zeros((2, F, padded, 32, 32), dtype=torch.int32, device="cuda")
for k in range(F):
    r.bind_color_tiles_pair(k, tiles[0, k].data_ptr(), tiles[1, k].data_ptr(), tiles[0, k].numel() * 4)
gathered = torch.zeros((F, padded, 32, 32), dtype=torch.int32, device="cuda")
lat, consumed, n = [], [[None] * F, [None] * F], 0
t0 = None
for it in range(64 + 600):
    if it == 64:
        r.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    k, par = n % F, (n // F) & 1
    if consumed[par][k] is not None and not consumed[par][k].query():
        consumed[par][k].synchronize()
    r.trace(); n += 1
    r.stream_wait_frame(xs.cuda_stream)
    if n % GB == 0:
        r.stream_wait_frame(ts.cuda_stream)          # timestamp of "the group's last frame is done"
        e_f = torch.cuda.Event(enable_timing=True); e_f.record(ts)
        k0 = (n - GB) % F
        with torch.cuda.stream(xs):
            gathered[k0:k0 + GB].copy_(tiles[par, k0:k0 + GB])
        e_x = torch.cuda.Event(enable_timing=True); e_x.record(xs)
        for j in range(k0, k0 + GB): consumed[par][j] = e_x
        if it >= 64: lat.append((e_f, e_x))
r.sync(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 600 * 1e6
ms = sorted(a.elapsed_time(b) * 1e3 for a, b in lat)
print(f"G={G} GB={GB} F={F}: {dt:.1f} us/frame; exchange finishes p10 {ms[len(ms)//10]:.0f} p50 {ms[len(ms)//2]:.0f} p90 {ms[len(ms)*9//10]:.0f} max {ms[-1]:.0f} us after its last frame")
