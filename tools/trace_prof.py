#!/usr/bin/env python3
"""What the iterations of the persistent per-ray tracer are made of, for the AO launch of config 5 (or --width/--height): a profiling build of libart is needed,
    make -C araytracingjourney_amd/csrc art_trace.o EXTRA=-DART_TRACE_PROF -B && hipcc --offload-arch=gfx950 -shared -fPIC -o araytracingjourney_amd/libart_prof.so araytracingjourney_amd/csrc/*.o -lz -ldl
    ART_LIB_PATH=$PWD/araytracingjourney_amd/libart_prof.so python tools/trace_prof.py"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:
    import torch  # noqa: F401
except Exception:
    pass
from araytracingjourney_amd import renderer, scenes, _lib
ap = argparse.ArgumentParser(); ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160); ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--tuning", default="")
a = ap.parse_args()
sc = scenes.sponza_like(1.0)
sc = scenes.Scene(sc.name, sc.primitives, sc.camera, scenes.sponza_lights(1))
tuning = {k: int(v) for k, v in (kv.split("=") for kv in a.tuning.split(",") if kv)} or None
r = renderer.renderer_for_scene(sc, (a.width, a.height), frames_in_flight=2, tuning=tuning)
L = _lib.load()
fn = C.CDLL(_lib.LIB_PATH).art_debug_trace_prof
out = (C.c_ulonglong * 8)()
r.render_frame(sync=False); r.trace_ao(a.spp); r.sync()
fn(out, 1)
r.render_frame(sync=False); r.trace_ao(a.spp); r.sync()
fn(out, 0)
it, nit, nl, lit, ll, rit, rl, act = [int(x) for x in out]
st = r.stats()
rays = st["ao_rays"]
print(json.dumps(dict(rays=rays, slots_refilled=rl, iterations_per_wave=it / 8192, node_iterations=nit, lanes_per_node_iteration=nl / max(nit, 1), triangle_iterations=lit, lanes_per_triangle_iteration=ll / max(lit, 1),
                      refill_iterations=rit, lanes_per_refill=rl / max(rit, 1), node_steps_per_ray=nl / max(rl, 1), triangle_tests_per_ray=ll / max(rl, 1), lanes_with_a_ray=act / max(it, 1)), indent=1))
