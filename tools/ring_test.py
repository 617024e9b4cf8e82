"""experiment: N contexts (frames in flight) round-robin vs one context; config 2"""
import sys, time
sys.path.insert(0, ".")
import torch
from araytracingjourney_amd import renderer, scenes
sc = scenes.sponza_like()
W, H = (1920, 1080) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
for nctx in (1, 2, 3, 4):
    rs = [renderer.renderer_for_scene(sc, (W, H)) for _ in range(nctx)]
    for r in rs:
        r.upload_state()
    for i in range(10):
        rs[i % nctx].trace()
    for r in rs:
        r.sync()
    K = 120
    t0 = time.perf_counter()
    for i in range(K):
        rs[i % nctx].trace()
    for r in rs:
        r.sync()
    dt = time.perf_counter() - t0
    st = rs[0].stats()
    rays = st["primary_rays"] + st["shadow_rays"]
    print(f"frames in flight {nctx}: {dt / K * 1e3:.4f} ms/frame, {rays * K / dt / 1e6:.0f} Mray/s", flush=True)
    for r in rs:
        r.close()
