#!/usr/bin/env python3
"""Where a wave of k_frame spends its life: shader-clock cycles per phase, summed over the waves of N frames.

Needs the profiling build of the library (the product build has no such symbol and this tool says so):

    make -C araytracingjourney_amd/csrc clean && make -C araytracingjourney_amd/csrc EXTRA=-DART_PHASE_PROF
    python tools/phase_prof.py [--width 1920 --height 1080 --lights 1 --frames 32]
    make -C araytracingjourney_amd/csrc clean && make -C araytracingjourney_amd/csrc          # back to the product build

The counters are s_memtime deltas taken by lane 0 at the phase boundaries (an s_waitcnt lgkmcnt(0) behind each: a boundary drains the wave's
scalar loads, nothing else) and stored per wave item (no atomics: 8 M of them on one line would be the profile) and say how a
wave's LIFETIME divides, not what the chip is busy with: 7 of 8 waves of a SIMD are waiting at any moment."""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from araytracingjourney_amd import _lib, scenes  # noqa: E402
from araytracingjourney_amd import renderer  # noqa: E402

PHASES = ["camera ray + ray_init", "primary packet walk", "surface (record, 3 textures, normal map) + depth/normal stores", "light: BRDF, radiance, shadow ray_init",
          "shadow packet walk", "accumulate + colour / bits stores"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--lights", type=int, default=1)
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--frames-in-flight", type=int, default=16)
    a = ap.parse_args()
    lib = _lib.load()
    try:
        fn = lib.art_debug_phase
    except AttributeError:
        sys.exit("libart.so is the product build: rebuild with EXTRA=-DART_PHASE_PROF (see the docstring)")
    fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int32], ctypes.c_int32
    sc = scenes.bistro_like() if a.scene == "bistro" else scenes.sponza_like()
    if a.scene != "bistro":
        sc = scenes.Scene(sc.name, sc.primitives, sc.camera, scenes.sponza_lights(a.lights))
    r = renderer.renderer_for_scene(sc, (a.width, a.height), frames_in_flight=a.frames_in_flight)
    r.upload_state()
    for _ in range(3 * a.frames_in_flight):     # the wave plan settles
        r.trace()
    r.sync()
    out = np.zeros((8, 1 << 18), dtype=np.uint32)
    assert fn(out.ctypes.data, 1) == 0
    for _ in range(a.frames):
        r.trace()
    r.sync()
    assert fn(out.ctypes.data, 1) == 0
    ran = out[6] != 0
    v = out[:6, ran].astype(np.float64)          # what the last frame to run each wave item measured
    waves, total = int(ran.sum()), v.sum()
    life = v.sum(0)
    print(f"{a.scene} {a.width}x{a.height}, {a.lights} light(s), {waves} waves a frame; a wave lives {total / waves:,.0f} clocks "
          f"(median {np.median(life):,.0f}, 90 % {np.percentile(life, 90):,.0f}, longest {life.max():,.0f})")
    for name, c in zip(PHASES, v.sum(1)):
        print(f"  {100 * c / total:5.1f} %  {c / waves:9,.0f} clocks a wave   {name}")


if __name__ == "__main__":
    main()
