#!/usr/bin/env python3
"""The beam walk (ArtTuning.packet_wide 3) against the per-ray box tests (1): every output of a frame must be the same bits; then frames per second of both.
    python tools/beam_probe.py [--scene bistro] [--width W --height H] [--lights N] [--frames K] [--detail D]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:
    import torch  # noqa: F401
except Exception:
    pass
from araytracingjourney_amd import renderer, scenes
ap = argparse.ArgumentParser(); ap.add_argument("--scene", default="sponza"); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--lights", type=int, default=1); ap.add_argument("--frames", type=int, default=400); ap.add_argument("--detail", type=float, default=1.0)
ap.add_argument("--tunings", default="packet_wide=1;packet_wide=3", help="ArtTuning settings to compare, separated by ;")
ap.add_argument("--no-compare", action="store_true")
a = ap.parse_args()
sc = scenes.bistro_like(a.detail) if a.scene == "bistro" else scenes.sponza_like(a.detail)
lights = sc.lights if a.scene == "bistro" else scenes.sponza_lights(a.lights)
sc = scenes.Scene(sc.name, sc.primitives, sc.camera, lights)
outs = {}
def parse(t): return {k: (float(v) if k in ("split_alpha", "refit_rebuild_ratio", "beam_fat") else int(v)) for k, v in (kv.split("=") for kv in t.split(",") if kv)}
for form in a.tunings.split(";"):
    if not a.no_compare:
        r = renderer.renderer_for_scene(sc, (a.width, a.height), keep_debug=True, frames_in_flight=1, tuning=parse(form))
        r.render_frame()
        tuv, ids = r.read_hits()
        outs[form] = (tuv.view(np.uint32).copy(), ids.copy(), r.read_shadow_bits().copy(), r.read_color().view(np.uint32).copy(), r.read_depth().view(np.uint32).copy(), r.read_normal().view(np.uint32).copy())
        r.close()
    r = renderer.renderer_for_scene(sc, (a.width, a.height), frames_in_flight=16, tuning=parse(form))
    for _ in range(200): r.render_frame(sync=False)
    r.sync()
    t0 = time.perf_counter()
    for _ in range(a.frames): r.render_frame(sync=False)
    r.sync()
    dt = time.perf_counter() - t0
    print(f"{form}: {dt / a.frames * 1e3:.4f} ms per frame over {a.frames} frames", flush=True)
    r.close()
forms = list(outs)
for f in forms[1:]:
    names = ("hit t/u/v", "hit ids", "shadow bits", "colour", "depth", "normal")
    for n, x, y in zip(names, outs[forms[0]], outs[f]):
        d = int((x != y).sum())
        print(f"{forms[0]} vs {f}: {n}: {'bit-equal' if d == 0 else str(d) + ' words differ'}")
