import sys, time
sys.path.insert(0, ".")
import torch
from araytracingjourney_amd import renderer, scenes
sc = scenes.sponza_like()
import os
FF = int(os.environ.get('PF', '16'))
GG = int(os.environ.get('PG', '32'))
for G, F, graph in ((GG, FF, True),):
    r = renderer.renderer_for_scene(sc, (1920, 1080), shard=(int(os.environ.get("PK", "0")), G), frames_in_flight=F)
    B = int(os.environ.get('PB', '1'))   # frames per launch
    if B > 1: r.set_frames_per_launch(B)
    r.upload_state(); r.set_graph_mode(graph)
    for i in range(40): r.trace()
    r.sync()
    K = 400
    t0 = time.perf_counter()
    for i in range(K): r.trace()
    t1 = time.perf_counter()
    r.sync()
    t2 = time.perf_counter()
    print(f"G={G} F={F} B={B}: issue {(t1-t0)/K/B*1e6:.1f} us/frame, total {(t2-t0)/K/B*1e6:.1f} us/frame", flush=True)
    r.close()
