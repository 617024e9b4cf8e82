#!/usr/bin/env python3
"""Wall time of a fenced burst of K frames (fence, K x art_trace, fence) against K, in one process: is the driver's 20-step figure a fixed cost of
filling and draining the ring, or a slower rate while the frames in flight are in step?   python tools/burst_sweep.py [--frames-in-flight 16] [--tuning k=v,...]"""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from araytracingjourney_amd import renderer, scenes  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--frames-in-flight", type=int, default=16); ap.add_argument("--tuning", default=""); ap.add_argument("--repeat", type=int, default=7)
ap.add_argument("--ks", default="1,2,4,8,16,20,24,32,48,64,96,128,256,512")
ap.add_argument("--idle-ms", type=float, default=0.0, help="host sleep between the fence and the burst")
a = ap.parse_args()
sc = scenes.sponza_like(); sc = scenes.Scene(sc.name, sc.primitives, sc.camera, scenes.sponza_lights(1))
tuning = {k: (float(v) if k == "split_alpha" else int(v)) for k, v in (kv.split("=") for kv in a.tuning.split(",") if kv)} or None
F = a.frames_in_flight
r = renderer.renderer_for_scene(sc, (1920, 1080), frames_in_flight=F, tuning=tuning)
r.upload_state()
t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    for _ in range(F): r.trace()
r.sync()
prev = None
for K in [int(x) for x in a.ks.split(",")]:
    ts = []
    for rep in range(a.repeat):
        for _ in range(5): r.trace()
        r.sync()
        if a.idle_ms: time.sleep(a.idle_ms / 1e3)
        t0 = time.perf_counter()
        for _ in range(K): r.trace()
        r.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    med = statistics.median(ts)
    print(f"K={K:4d}  median {med:8.3f} ms  min {min(ts):8.3f}  per frame {med / K * 1e3:7.1f} us" + (f"  marginal since K={prev[0]}: {(med - prev[1]) / (K - prev[0]) * 1e3:6.1f} us/frame" if prev else ""), flush=True)
    prev = (K, med)
r.close()
