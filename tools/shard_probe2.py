import sys, time
sys.path.insert(0, ".")
import torch
from araytracingjourney_amd import renderer, scenes
G, F = int(sys.argv[1]), int(sys.argv[2])
sc = scenes.sponza_like()
r = renderer.renderer_for_scene(sc, (1920, 1080), shard=(0, G), frames_in_flight=F)
r.upload_state(); r.set_graph_mode(len(sys.argv) > 3)
for i in range(20): r.trace()
r.sync()
K = 200
t0 = time.perf_counter()
for i in range(K): r.trace()
r.sync()
print("ms/frame", (time.perf_counter() - t0) / K * 1e3)
