#!/usr/bin/env python3
"""art_scene_build (+ the 4-wide collapse) of a bench scene, a few times over: device time per build (ArtStats.build_ms) and wall clock.
    python tools/build_probe.py [--scene sponza|bistro] [--n 5] [--tuning k=v,...]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:
    import torch  # noqa: F401
except Exception:
    pass
from araytracingjourney_amd import renderer, scenes
ap = argparse.ArgumentParser(); ap.add_argument("--scene", default="sponza"); ap.add_argument("--n", type=int, default=5); ap.add_argument("--tuning", default="")
a = ap.parse_args()
sc = scenes.bistro_like(1.0) if a.scene == "bistro" else scenes.sponza_like(1.0)
tuning = {k: int(v) for k, v in (kv.split("=") for kv in a.tuning.split(",") if kv)} or None
r = renderer.renderer_for_scene(sc, (640, 360), tuning=tuning)
first = round(r.stats()["build_ms"], 2)   # the first build of the process
r.render_frame()
out = []
for _ in range(a.n):
    t0 = time.perf_counter()
    r.prepare_first_frame(); r.render_frame()          # build, then a frame (the collapse is made on first use)
    out.append((round(r.stats()["build_ms"], 2), round((time.perf_counter() - t0) * 1e3, 2)))
print(json.dumps(dict(scene=a.scene, triangles=sc.n_tris, first_build_ms_device=first, build_ms_device_and_wall=out)))
