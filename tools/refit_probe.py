#!/usr/bin/env python3
"""A model of the bench scene moved before every frame (art_scene_set_model_matrix): refit and frame times for a given ring depth and number of versions.
    python tools/refit_probe.py [--scene sponza|bistro] [--frames-in-flight 16] [--versions 3] [--steps 200] [--mover -1]
Under rocprofv3 --kernel-trace --stats it shows what a refit is made of (k_retri, k_wide_refit per level, k_wide_refit_top, k_wide_cost)."""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
try:
    import torch  # noqa: F401
except Exception:
    pass
from araytracingjourney_amd import renderer, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sponza"); ap.add_argument("--frames-in-flight", type=int, default=16); ap.add_argument("--versions", type=int, default=3)
ap.add_argument("--steps", type=int, default=200); ap.add_argument("--mover", type=int, default=-1); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
sc = scenes.bistro_like(1.0) if a.scene == "bistro" else scenes.sponza_like(1.0)
lights = sc.lights if a.scene == "bistro" else scenes.sponza_lights(1)
j = a.mover % len(sc.primitives)
r = renderer.Renderer((a.width, a.height), frames_in_flight=a.frames_in_flight, tuning={"as_versions": a.versions, "refit_rebuild_ratio": -1.0, "log": 1})
r.add_model([p for i, p in enumerate(sc.primitives) if i != j]); r.add_model([sc.primitives[j]])
cam = r.camera_mut()
cam.set_pos(sc.camera["pos"]); cam.set_dir(sc.camera["dir"]); cam.set_fovy(sc.camera["fovy"]); cam.set_znear(sc.camera["znear"]); cam.set_zfar(sc.camera["zfar"])
for d in lights:
    r.lights_mut().push_dict(d)
r.prepare_first_frame(); r.upload_state()
base = np.vstack([np.asarray(sc.primitives[j].model, np.float64).reshape(3, 4), [0, 0, 0, 1]])
poses = []
for i in range(8):
    an = 2 * math.pi * i / 8
    ry = np.array([[math.cos(an), 0, math.sin(an), 0], [0, 1, 0, 0], [-math.sin(an), 0, math.cos(an), 0], [0, 0, 0, 1]])
    t = np.eye(4); t[:3, 3] = (0.15 * math.cos(an) - 0.15, 0.05 * math.sin(2 * an), 0.15 * math.sin(an))
    poses.append(np.ascontiguousarray((t @ ry @ base)[:3], np.float32))
model = r.models_mut()[1]
for _ in range(3 * a.frames_in_flight):
    r.trace()
r.sync()
alone = []
for m in poses:
    model.set_model_matrix(m); r.trace(); r.sync(); alone.append(r.stats()["refit_ms"])
t0 = time.perf_counter()
for i in range(a.steps):
    model.set_model_matrix(poses[i % 8]); r.trace()
r.sync()
dt = time.perf_counter() - t0
t1 = time.perf_counter()
for i in range(a.steps):
    r.trace()
r.sync()
ds = time.perf_counter() - t1
st = r.stats()
print(json.dumps(dict(scene=a.scene, triangles=sc.n_tris, moving_triangles=sc.primitives[j].n_tris, frames_in_flight=a.frames_in_flight, versions=a.versions, refit_ms_alone=sorted(alone)[4],
                      ms_per_moving_frame=dt * 1e3 / a.steps, ms_per_static_frame=ds * 1e3 / a.steps, refits=st["refits"], cost_ratio=st["refit_cost_ratio"], build_ms=st["build_ms"])))
