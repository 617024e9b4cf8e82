"""diagnostic build only (libart_diag.so): lane 0 of every wave leaves its start/end wall-clock ticks in the colour of its pixel"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import numpy as np
from araytracingjourney_amd import renderer, scenes
k = int(sys.argv[1])
sc = scenes.sponza_like()
r = renderer.renderer_for_scene(sc, (1920, 1080), shard=(k, 8), frames_in_flight=16)
r.upload_state()
for i in range(200): r.trace()
r.sync()
c = r.read_color().view(np.uint32)          # the last frame's log
lane0 = c[0::8, 0::8]                        # one 8x8 block per entry
t0 = lane0[..., 0].astype(np.uint64) | (lane0[..., 1].astype(np.uint64) << 32)
t1 = lane0[..., 2].astype(np.uint64) | (lane0[..., 3].astype(np.uint64) << 32)
m = (t1 > t0) & (t0 > 0)
t0, t1 = t0[m].astype(np.int64), t1[m].astype(np.int64)
base = t0.min()
dur = (t1 - t0) / 100.0                      # us at 100 MHz
print(f"shard {k}: {m.sum()} waves, launch span {(t1.max() - base) / 100.0:.0f} us; wave duration mean {dur.mean():.1f} p50 {np.percentile(dur,50):.1f} p90 {np.percentile(dur,90):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f}")
start = (t0 - base) / 100.0; end = (t1 - base) / 100.0
print("start times p50 %.0f p90 %.0f max %.0f; end times p50 %.0f p90 %.0f p99 %.0f max %.0f" % (np.percentile(start,50), np.percentile(start,90), start.max(), np.percentile(end,50), np.percentile(end,90), np.percentile(end,99), end.max()))
order = np.argsort(-dur)[:8]
print("longest waves (start, dur):", [(round(float(start[i])), round(float(dur[i]))) for i in order])
