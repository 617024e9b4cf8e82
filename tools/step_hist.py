#!/usr/bin/env python3
"""How the packet steps of a fused frame are spread over its 8x8 blocks (every block one wave: the wave plan is switched off):
   tools/step_hist.py [sponza|bistro] [W H] [lights]
percentiles of the steps per wave, and for a row of budgets B: the blocks above B, the share of ALL steps their waves make, and the share that lies above B
(what a walk that stops being a packet walk at B steps would not make as a packet)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import numpy as np
from araytracingjourney_amd import renderer, scenes
from araytracingjourney_amd._lib import check

scene = sys.argv[1] if len(sys.argv) > 1 else "sponza"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
nl = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if scene == "bistro":
    sc = scenes.bistro_like(1.0)
else:
    sc = scenes.sponza_like(1.0)
    sc = scenes.Scene(sc.name, sc.primitives, sc.camera, scenes.sponza_lights(nl))
r = renderer.renderer_for_scene(sc, (W, H), frames_in_flight=1, tuning=dict(fixed_waves=1))
r.upload_state()


def sample():
    cap = (W // 8 + 8) * (H // 8 + 8) * 2
    items = np.zeros((cap, 2), np.uint32); steps = np.zeros(cap, np.uint32); n = C.c_uint32()
    check(r._L.art_sample_wave_steps(r._ctx, items.ctypes.data, steps.ctypes.data, cap, C.byref(n)))
    m = items[:n.value, 1] != 0
    return steps[:n.value][m].astype(np.int64)


def report(tag, s):
    tot = s.sum()
    print(f"{tag}: {len(s)} waves, {tot} steps, mean {s.mean():.1f}  p50 {np.percentile(s, 50):.0f} p90 {np.percentile(s, 90):.0f} p99 {np.percentile(s, 99):.0f} p99.9 {np.percentile(s, 99.9):.0f} max {s.max()}")
    for B in (64, 96, 128, 192, 256, 384, 512, 1024):
        over = s > B
        print(f"   budget {B:5d}: {over.sum():6d} blocks ({100.0 * over.mean():5.2f} %) above, their waves make {100.0 * s[over].sum() / tot:5.1f} % of all steps, {100.0 * (s[over] - B).sum() / tot:5.1f} % lie above the budget")


report(f"{scene} {W}x{H} {nl} light(s), the scene's camera", sample())
if scene == "sponza":
    for i, p in enumerate(scenes.camera_path(sc, 8)):
        r._camera = renderer.Camera(p["pos"], p["dir"], W / H, p["fovy"], p["znear"], p["zfar"])
        r.upload_state()
        if i in (2, 5):
            report(f"camera path pose {i}", sample())
