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
r = renderer.renderer_for_scene(sc, (W, H), shard=(int(os.environ.get("PK", "0")), G), frames_in_flight=F, packed_tiles=True)
PB = int(os.environ.get('PB', '1'))   # frames per launch
if PB > 1: r.set_frames_per_launch(PB)
r.upload_state()
PRIO = -1 if os.environ.get('HIPRIO', '1') == '1' else 0
XS = int(os.environ.get('XS', '1'))   # exchange streams used in turn (one exchange = one gather group)
streams = [torch.cuda.Stream(priority=PRIO) for _ in range(XS)]
stream = streams[0]; torch.cuda.set_stream(stream)
group_no = [0]
owned, padded = r.shard_tile_count()
PAIR = os.environ.get('PAIR', '1') == '1'
NB = int(os.environ.get('NB', '2')) if PAIR else 1   # tile buffers per ring slot
tiles = torch.zeros((NB, F, PB, padded, 32, 32), dtype=torch.int32, device="cuda")
for k in range(F):
    if PAIR: r.bind_color_tiles_ring(k, [tiles[b, k].data_ptr() for b in range(NB)], tiles[0, k].numel() * 4)
    else: r.bind_color_tiles(k, tiles[0, k].data_ptr(), tiles[0, k].numel() * 4)
gathered = torch.zeros((G, F, PB, padded, 32, 32), dtype=torch.int32, device="cuda")
frames = [torch.zeros((GB * PB, H, W), dtype=torch.int32, device="cuda") for _ in range(XS)]
consumed = [[None] * F for _ in range(NB)]
pending = [0, 0]
frame_no = [0]
HOSTX = os.environ.get('HOSTX', '0') == '1'   # the exchange is submitted once the host sees its frames done (no device-side waits)
fifo = []
def exchange(force=False, form=True):
    k0, n = pending
    if n and form:
        pending[0], pending[1] = (k0 + n) % F, 0
        par = ((frame_no[0] - 1) // F) % NB
        fifo.append((k0, n, par, frame_no[0] - n))
        for j in range(k0, k0 + n): consumed[par][j] = 'queued'   # not exchanged yet: the buffer must not be rewritten
    while fifo:
        k0, n, par, first = fifo[0]
        if HOSTX:
            tq = time.perf_counter(); ok = r.frames_done(first, n); prof['poll'] += time.perf_counter() - tq; prof['polls'] += 1
            if not ok:
                if not force: return
                r.sync()
        fifo.pop(0)
        run_exchange(k0, n, par)
def run_exchange(k0, n, par):
    xs = streams[group_no[0] % XS]; group_no[0] += 1
    if MODE == "noexchange":
        for j in range(k0, k0 + n): consumed[par][j] = None
        return
    e0 = torch.cuda.Event(enable_timing=True); e0.record(xs)
    with torch.cuda.stream(xs):
        if MODE != "nocopy": gathered[0, k0:k0 + n].copy_(tiles[par, k0:k0 + n])
        e1 = torch.cuda.Event(enable_timing=True); e1.record(xs)
        if MODE != "nountile":
            r.untile_gathered(gathered[0, k0].data_ptr(), G, frames[(group_no[0] - 1) % XS].data_ptr(), xs.cuda_stream, shard_stride_tiles=F * PB * padded, n_frames=n * PB)
    ev = torch.cuda.Event(enable_timing=True); ev.record(xs)
    xlog.append((e0, e1, ev))
    for j in range(k0, k0 + n): consumed[par][j] = ev
def step():
    _, k = r.frames_in_flight()
    par = (frame_no[0] // F) % NB
    while consumed[par][k] == 'queued': exchange(form=False)   # (spins on the host until the group's frames are done)
    if consumed[par][k] is not None and MODE != "nowait":
        if os.environ.get('HOSTWAIT', '0') == '1':
            tq = time.perf_counter()
            if not consumed[par][k].query():
                consumed[par][k].synchronize(); prof['waits'] += 1
            prof['gate'] += time.perf_counter() - tq
        else: r.wait_external_event(consumed[par][k].cuda_event)
    tq = time.perf_counter()
    r.trace()
    prof['trace'] += time.perf_counter() - tq
    frame_no[0] += 1
    tq = time.perf_counter()
    if not HOSTX: r.stream_wait_frame(streams[group_no[0] % XS].cuda_stream)
    prof['swf'] += time.perf_counter() - tq
    pending[1] += 1
    tq = time.perf_counter()
    if pending[1] == GB or k + 1 == F: exchange()
    elif HOSTX and fifo and frame_no[0] % int(os.environ.get('POLL', '1')) == 0: exchange(form=False)
    prof['xchg'] += time.perf_counter() - tq
xlog = []
prof = dict(gate=0.0, trace=0.0, swf=0.0, xchg=0.0, waits=0, poll=0.0, polls=0)
for _ in range(64): step()
exchange(True); r.sync(); torch.cuda.synchronize()
K = int(os.environ.get("PKN", "800"))   # launches timed
xlog.clear()
prof = dict(gate=0.0, trace=0.0, swf=0.0, xchg=0.0, waits=0, poll=0.0, polls=0)
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
exchange(True); r.sync(); torch.cuda.synchronize()
t2 = time.perf_counter()
print({k: (round(v / K * 1e6, 1) if k not in ("waits", "polls") else v) for k, v in prof.items()}, "us/frame")
if xlog:
    med = lambda v: sorted(v)[len(v) // 2]
    print("exchange kernels, median us: copy", round(med([a.elapsed_time(b) for a, b, c in xlog]) * 1e3), "un-tile", round(med([b.elapsed_time(c) for a, b, c in xlog]) * 1e3),
          "| from one exchange's end to the next", round(med([xlog[i][2].elapsed_time(xlog[i + 1][2]) for i in range(len(xlog) - 1)]) * 1e3))
print(f"G={G} GB={GB} B={PB} {MODE}: host loop {(t1-t0)/K/PB*1e6:.1f} us/frame, pipeline {(t2-t0)/K/PB*1e6:.1f} us/frame", flush=True)
