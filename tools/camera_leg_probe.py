#!/usr/bin/env python3
"""The driver's protocol on a camera that stands still, on one that moves every frame (bench.py's closed path of 8 poses) and on one that creeps like the
reference's own (main.rs:69-131: 0.002 units per millisecond of frame time, a few mouse counts of 0.002 rad): fenced bursts of K frames through a ring of F.
   tools/camera_leg_probe.py [K=20] [F=3] [reps=6] [tuning k=v,...]"""
import os, sys, time, math
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from araytracingjourney_amd import renderer, scenes

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
tuning = {k: (float(v) if "." in v else int(v)) for k, v in (kv.split("=") for kv in (sys.argv[4] if len(sys.argv) > 4 else "").split(",") if kv)}
tuning.setdefault("hw_queues", 16)
W, H = 1920, 1080
sc = scenes.sponza_like(1.0)
sc = scenes.Scene(sc.name, sc.primitives, sc.camera, scenes.sponza_lights(1))
r = renderer.renderer_for_scene(sc, (W, H), frames_in_flight=F, tuning=tuning)
r.upload_state()
cam0 = renderer.Camera(sc.camera["pos"], sc.camera["dir"], W / H, sc.camera["fovy"], sc.camera["znear"], sc.camera["zfar"])
path8 = [renderer.Camera(p["pos"], p["dir"], W / H, p["fovy"], p["znear"], p["zfar"]) for p in scenes.camera_path(sc, 8)]
# the reference's camera: 0.002 units per ms -> at 0.17 ms per frame 0.00034 units; a creep a hundred times faster than that still moves the heavy blocks by a pixel or so
p0, d0 = np.asarray(sc.camera["pos"], np.float64), np.asarray(sc.camera["dir"], np.float64)
creep = []
for i in range(64):
    a = 2 * math.pi * i / 64
    creep.append(renderer.Camera(tuple(p0 + np.array([0.03 * math.sin(a), 0.0, 0.03 * (1 - math.cos(a))])), tuple(d0 + np.array([0.0, 0.01 * math.sin(a), 0.02 * math.sin(a)])), W / H,
                                 sc.camera["fovy"], sc.camera["znear"], sc.camera["zfar"]))


host = dict(up=[], tr=[])


def burst(cams, n):
    for i in range(n):
        t0 = time.perf_counter()
        if cams is not None:
            r._camera = cams[i % len(cams)]
            r.upload_state()
        t1 = time.perf_counter()
        r.trace()
        t2 = time.perf_counter()
        host["up"].append(t1 - t0); host["tr"].append(t2 - t1)


def leg(name, cams):
    for _ in range(3 * F):
        burst(cams, 1)
    r.sync()
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:      # settle like bench.py
        burst(cams, 3 * F)
    r.sync()
    out = []
    host["up"].clear(); host["tr"].clear()
    for _ in range(reps):
        burst(cams, 5); r.sync()
        t0 = time.perf_counter()
        burst(cams, K); r.sync()
        out.append((time.perf_counter() - t0) * 1e3 / K)
    st = r.stats()
    up, tr = np.array(host["up"]) * 1e6, np.array(host["tr"]) * 1e6
    print(f"   host us per frame: upload_state mean {up.mean():.1f} max {up.max():.0f}; art_trace mean {tr.mean():.1f} p90 {np.percentile(tr, 90):.0f} max {tr.max():.0f}; calls of art_trace above 50 us: {(tr > 50).sum()} of {len(tr)}")
    print(f"{name:28s} ms/frame " + " ".join(f"{x:.4f}" for x in out) + f"   median {sorted(out)[len(out) // 2]:.4f}  split_blocks {st['split_blocks']}", flush=True)


r._camera = cam0; r.upload_state()
leg("static", None)
leg("path of 8 poses", path8)
leg("creeping camera (2 px/frame)", creep)
walk = [renderer.Camera(p["pos"], p["dir"], W / H, p["fovy"], p["znear"], p["zfar"]) for p in scenes.camera_walk(sc, 64)]
leg("the reference's walk", walk)
r._camera = cam0; r.upload_state()
leg("static again", None)
