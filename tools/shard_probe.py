"""experiment: how fast does ONE GPU render its share of an N-way sharded frame (no gather)? upper bound for strong scaling"""
import sys, time
sys.path.insert(0, ".")
import torch
from araytracingjourney_amd import renderer, scenes
sc = scenes.sponza_like()
W, H = 1920, 1080
base = None
import os
for G in (1, 4, 8):
    for F, graph in ((4, True), (8, True), (12, True), (16, True)):
        r = renderer.renderer_for_scene(sc, (W, H), shard=(0, G), frames_in_flight=F)
        r.upload_state(); r.set_graph_mode(graph)
        for i in range(20): r.trace()
        r.sync()
        K = 300
        t0 = time.perf_counter()
        for i in range(K): r.trace()
        r.sync()
        dt = (time.perf_counter() - t0) / K
        if base is None: base = dt
        print(f"shard 0 of {G}, F={F} graph={graph}: {dt*1e3:.4f} ms/frame -> ideal speedup {base/dt:.2f}x", flush=True)
        r.close()
