"""rank 0's per-frame pipeline of bench.py for N > 1, without the network: shard 0 of G, the exchange replaced by a device copy of
its own tiles into the gather buffer, then the un-tile of every frame.  Shows whether the host loop keeps up with a 1/G share."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import torch
from araytracingjourney_amd import renderer, scenes
G, F, GB = int(sys.argv[1]), int(os.environ.get("PF", "16")), int(sys.argv[2]) if len(sys.argv) > 2 else 4
MODE = sys.argv[3] if len(sys.argv) > 3 else "full"   # full | nountile | nocopy | noexchange | nowait
sc = scenes.sponza_like()
W, H = 1920, 1080
r = renderer.renderer_for_scene(sc, (W, H), shard=(0, G), frames_in_flight=F, packed_tiles=True)
r.upload_state()
PRIO = -1 if os.environ.get('HIPRIO', '1') == '1' else 0
stream = torch.cuda.Stream(priority=PRIO); torch.cuda.set_stream(stream)
owned, padded = r.shard_tile_count()
PAIR = os.environ.get('PAIR', '1') == '1'
tiles = torch.zeros((2, F, padded, 32, 32), dtype=torch.int32, device="cuda")
for k in range(F):
    if PAIR: r.bind_color_tiles_pair(k, tiles[0, k].data_ptr(), tiles[1, k].data_ptr(), tiles[0, k].numel() * 4)
    else: r.bind_color_tiles(k, tiles[0, k].data_ptr(), tiles[0, k].numel() * 4)
gathered = torch.zeros((G, F, padded, 32, 32), dtype=torch.int32, device="cuda")
frame = torch.zeros((GB, H, W), dtype=torch.int32, device="cuda")
consumed = [[None] * F, [None] * F]
pending = [0, 0]
frame_no = [0]
def exchange():
    k0, n = pending
    if n == 0: return
    pending[0], pending[1] = (k0 + n) % F, 0
    par = ((frame_no[0] - 1) // F) & 1 if PAIR else 0
    if MODE == "noexchange": return
    if MODE != "nocopy": gathered[0, k0:k0 + n].copy_(tiles[par, k0:k0 + n])
    if MODE != "nountile":
        r.untile_gathered(gathered[0, k0].data_ptr(), G, frame.data_ptr(), stream.cuda_stream, shard_stride_tiles=F * padded, n_frames=n)
    ev = torch.cuda.Event(); ev.record(stream)
    for j in range(k0, k0 + n): consumed[par][j] = ev
def step():
    _, k = r.frames_in_flight()
    par = (frame_no[0] // F) & 1 if PAIR else 0
    if consumed[par][k] is not None and MODE != "nowait":
        if os.environ.get('HOSTWAIT', '0') == '1':
            if not consumed[par][k].query(): consumed[par][k].synchronize()
        else: r.wait_external_event(consumed[par][k].cuda_event)
    r.trace()
    frame_no[0] += 1
    r.stream_wait_frame(stream.cuda_stream)
    pending[1] += 1
    if pending[1] == GB or k + 1 == F: exchange()
for _ in range(64): step()
exchange(); r.sync(); torch.cuda.synchronize()
K = 800
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
exchange(); r.sync(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"G={G} GB={GB} {MODE}: host loop {(t1-t0)/K*1e6:.1f} us/frame, pipeline {(t2-t0)/K*1e6:.1f} us/frame", flush=True)
