"""does a kernel launch cost the launches already running something?  The whole frame, 16 in flight, while a side stream starts N tiny kernels per frame."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "18")
sys.path.insert(0, ".")
import torch
from araytracingjourney_amd import renderer, scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sc = scenes.sponza_like()
r = renderer.renderer_for_scene(sc, (1920, 1080), frames_in_flight=16)
r.upload_state()
side = torch.cuda.Stream()
x = torch.zeros(64, device="cuda")
for i in range(64): r.trace()
r.sync()
K = 600
t0 = time.perf_counter()
for i in range(K):
    r.trace()
    with torch.cuda.stream(side):
        for _ in range(N): x.add_(1.0)
r.sync(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K * 1e6
print(f"{N} tiny launches per frame beside it: {dt:.1f} us per frame")
