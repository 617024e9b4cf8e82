#!/usr/bin/env python3
"""What the packet walks of the fused frame are made of (node steps, triangle steps, how many of them find a hit ...), per BASELINE config; needs a profiling build:
    make -C araytracingjourney_amd/csrc art_trace.o EXTRA=-DART_PACKET_PROF -B && hipcc --offload-arch=gfx950 -shared -fPIC -o araytracingjourney_amd/libart_prof.so araytracingjourney_amd/csrc/*.o -lz -ldl
    ART_LIB_PATH=$PWD/araytracingjourney_amd/libart_prof.so python tools/packet_prof.py [--scene bistro] [--width W --height H --lights N] [--tuning k=v,...]"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:
    import torch  # noqa: F401
except Exception:
    pass
from araytracingjourney_amd import renderer, scenes, _lib
ap = argparse.ArgumentParser(); ap.add_argument("--scene", default="sponza"); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--lights", type=int, default=1); ap.add_argument("--tuning", default="")
a = ap.parse_args()
sc = scenes.bistro_like(1.0) if a.scene == "bistro" else scenes.sponza_like(1.0)
lights = sc.lights if a.scene == "bistro" else scenes.sponza_lights(a.lights)
sc = scenes.Scene(sc.name, sc.primitives, sc.camera, lights)
tuning = {k: (float(v) if k in ("split_alpha", "refit_rebuild_ratio") else int(v)) for k, v in (kv.split("=") for kv in a.tuning.split(",") if kv)} or None
r = renderer.renderer_for_scene(sc, (a.width, a.height), fixed_waves=True, tuning=tuning)
fn = C.CDLL(_lib.LIB_PATH).art_debug_packet_prof
out = (C.c_ulonglong * 24)()
r.render_frame(); fn(out, 1); r.render_frame(); fn(out, 0)
names = ("walks", "node_steps", "triangle_steps", "triangle_steps_from_the_stack", "triangle_steps_with_a_hit", "lanes_that_hit", "child_boxes_hit", "mixed_octant_walks_and_fat_beams", "cycles_in_node_steps", "cycles_in_triangle_steps", "cycles_in_beam_setup", "spare")
res = {}
for k, base in (("primary", 0), ("shadow", 12)):
    v = dict(zip(names, [int(x) for x in out[base:base + 12]]))
    w = max(v["walks"], 1)
    res[k] = dict(v, node_steps_per_walk=v["node_steps"] / w, triangle_steps_per_walk=v["triangle_steps"] / w, child_boxes_hit_per_node_step=v["child_boxes_hit"] / max(v["node_steps"], 1),
                  share_of_triangle_steps_with_a_hit=v["triangle_steps_with_a_hit"] / max(v["triangle_steps"], 1), share_from_the_stack=v["triangle_steps_from_the_stack"] / max(v["triangle_steps"], 1),
                  cycles_per_node_step=v["cycles_in_node_steps"] / max(v["node_steps"], 1), cycles_per_triangle_step=v["cycles_in_triangle_steps"] / max(v["triangle_steps"], 1), setup_cycles_per_walk=v["cycles_in_beam_setup"] / w)
print(json.dumps(res, indent=1))
