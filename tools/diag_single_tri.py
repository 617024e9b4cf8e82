import sys, numpy as np
sys.path.insert(0, '.')
import torch
from araytracingjourney_amd import scenes, renderer
mb = scenes.MeshBuilder()
mb.add([(0, 0, 2), (1, 0, 2), (0, 1, 2)], [(0, 0), (1, 0), (0, 1)], [(0, 0, -1)] * 3, [(1, 0, 0, 1)] * 3, [0, 1, 2])
sc = scenes.Scene("tri", [mb.finish(scenes.constant_texture((200, 200, 200)))], scenes.cornell().camera, [])
r = renderer.renderer_for_scene(sc, (8, 8))
print("built", r.stats(), flush=True)
rays = np.array([[0.25, 0.25, 0, 0.001, 0, 0, 1, 100]], np.float32)
print(r.query_closest(rays), flush=True)
