#!/usr/bin/env python3
"""What the frame rate would be if the boundary handed the outputs back in host buffers every frame: one frame traced, then
art_read_color / depth / normal (pageable host memory, as a caller's plain buffers are).  Never bench.py's `value`: the contract's inputs and
outputs stay resident in HBM."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from araytracingjourney_amd import renderer, scenes
sc = scenes.sponza_like()
sc = scenes.Scene(sc.name, sc.primitives, sc.camera, scenes.sponza_lights(1))
r = renderer.renderer_for_scene(sc, (1920, 1080))
r.upload_state()
for _ in range(5):
    r.trace(); r.sync()
st = r.stats(); rays = st["primary_rays"] + st["shadow_rays"]
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    r.trace(); r.sync()
    c = r.read_color(); d = r.read_depth(); n = r.read_normal()
    ts.append(time.perf_counter() - t0)
ts.sort()
nbytes = c.nbytes + d.nbytes + n.nbytes
print(f"frame + read-back of colour, depth, normal ({nbytes/1e6:.1f} MB): median {ts[10]*1e3:.3f} ms = {rays/ts[10]/1e6:,.0f} Mray/s")
t = []
for _ in range(20):
    t0 = time.perf_counter(); c = r.read_color(); d = r.read_depth(); n = r.read_normal(); t.append(time.perf_counter() - t0)
t.sort()
print(f"read-back alone: median {t[10]*1e3:.3f} ms = {nbytes/t[10]/1e9:.1f} GB/s")
