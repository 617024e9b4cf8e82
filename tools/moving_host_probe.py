#!/usr/bin/env python3
"""Host time of the two calls a moving-model frame makes (set_model_matrix, trace) while frames are in flight, config 2's scene:
   tools/moving_host_probe.py [frames=3000] [tuning k=v,...] [moving primitive=24]
(primitive 24 is the 164 k-triangle model of bench.py's leg; 2 has 3 072 triangles: the same ring of versions behind a refit of next to nothing)"""
import os, sys, time, math
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from araytracingjourney_amd import renderer, scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
tuning = {k: (float(v) if "." in v else int(v)) for k, v in (kv.split("=") for kv in (sys.argv[2] if len(sys.argv) > 2 else "").split(",") if kv)}
tuning.setdefault("hw_queues", 16)
MOVER = int(sys.argv[3]) if len(sys.argv) > 3 else 24
W, H = 1920, 1080
sc = scenes.sponza_like(1.0)
mv = renderer.Renderer((W, H), frames_in_flight=8, tuning=tuning, dynamic_scene=True)
mv.add_model(sc.primitives[:MOVER] + sc.primitives[MOVER + 1:]); mv.add_model(sc.primitives[MOVER:MOVER + 1])
cam = mv.camera_mut()
cam.set_pos(sc.camera["pos"]); cam.set_dir(sc.camera["dir"]); cam.set_fovy(sc.camera["fovy"]); cam.set_znear(sc.camera["znear"]); cam.set_zfar(sc.camera["zfar"])
for d in scenes.sponza_lights(1):
    mv.lights_mut().push_dict(d)
mv.prepare_first_frame(); mv.upload_state()
base = np.vstack([np.asarray(sc.primitives[MOVER].model, np.float64).reshape(3, 4), [0, 0, 0, 1]])
poses = []
for i in range(64):
    k = min(i, 64 - i); a = 2.0 / 6000.0 * k
    ry = np.array([[math.cos(a), 0, math.sin(a), 0], [0, 1, 0, 0], [-math.sin(a), 0, math.cos(a), 0], [0, 0, 0, 1]])
    t = np.eye(4); t[:3, 3] = (2.0 / 6000.0 * k, 0.0, 0.0)
    poses.append(np.ascontiguousarray((t @ ry @ base)[:3], np.float32))
model = mv.models_mut()[1]
for i in range(200):
    model.set_model_matrix(poses[i % 64]); mv.trace()
mv.sync()
a, b = np.zeros(N), np.zeros(N)
t0 = time.perf_counter()
for i in range(N):
    t1 = time.perf_counter()
    model.set_model_matrix(poses[i % 64])
    t2 = time.perf_counter()
    mv.trace()
    t3 = time.perf_counter()
    a[i] = t2 - t1; b[i] = t3 - t2
t_issue = time.perf_counter() - t0
mv.sync()
t_all = time.perf_counter() - t0
for name, v in (("set_model_matrix", a), ("trace", b)):
    v = v * 1e6
    print(f"{name:18s} mean {v.mean():7.1f} us  p50 {np.percentile(v, 50):7.1f}  p90 {np.percentile(v, 90):7.1f}  p99 {np.percentile(v, 99):7.1f}  max {v.max():8.1f}")
print(f"issued in {t_issue * 1e3 / N:.4f} ms a frame, done in {t_all * 1e3 / N:.4f} ms a frame; tuning {tuning}; primitive {MOVER} moves ({len(sc.primitives[MOVER].indices) // 3} triangles); refit_ms {mv.stats()['refit_ms']:.4f}")
