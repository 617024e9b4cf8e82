"""renders config 2 through the whole chain (trace + AO + present) and saves a downscaled PNG of the presented frame"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
from PIL import Image
from araytracingjourney_amd import renderer, scenes
sc = scenes.sponza_like()
sc = scenes.Scene(sc.name, sc.primitives, sc.camera, scenes.sponza_lights(4))
r = renderer.renderer_for_scene(sc, (1920, 1080))
r.render_frame(sync=False)
r.trace_ao(16)
r.present()
bgra = r.read_present()
Image.fromarray(bgra[..., [2, 1, 0]]).resize((640, 360), Image.LANCZOS).save("gpurun_out/sponza_like_presented.png")
Image.fromarray(r.read_ao().astype(np.uint8)).resize((640, 360), Image.LANCZOS).save("gpurun_out/sponza_like_ao.png")
print(r.stats())
