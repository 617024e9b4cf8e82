#!/usr/bin/env python3
"""What art_create and the first art_scene_build of a process cost (the code objects are loaded there), and the size of libart.so: tools/create_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
t0 = time.perf_counter()
from araytracingjourney_amd import renderer, scenes, _lib
t1 = time.perf_counter()
sc = scenes.sponza_like(0.12)
t2 = time.perf_counter()
r = renderer.Renderer((640, 360))
t3 = time.perf_counter()
r2 = renderer.Renderer((640, 360))
t4 = time.perf_counter()
r2.close()
rr = renderer.renderer_for_scene(sc, (640, 360))
t5 = time.perf_counter()
rr.render_frame()
t6 = time.perf_counter()
print(f"libart.so {os.path.getsize(_lib.LIB_PATH) / 1e6:.2f} MB; first art_create of the process {1e3 * (t3 - t2):.1f} ms, second {1e3 * (t4 - t3):.1f} ms; context + scene build (31 k triangles) {1e3 * (t5 - t4):.1f} ms; first frame {1e3 * (t6 - t5):.1f} ms")
