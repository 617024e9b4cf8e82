"""does a context created after another one was destroyed run as fast as the first? (shard 3 of 8, 16 frames in flight)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
from araytracingjourney_amd import renderer, scenes
sc = scenes.sponza_like()
def run():
    r = renderer.renderer_for_scene(sc, (1920, 1080), shard=(3, 8), frames_in_flight=16)
    r.upload_state()
    for i in range(40): r.trace()
    r.sync()
    t0 = time.perf_counter()
    for i in range(400): r.trace()
    r.sync()
    dt = (time.perf_counter() - t0) / 400 * 1e6
    r.close()
    return dt
print("first context %.1f us/frame, second %.1f, third %.1f" % (run(), run(), run()))
w = renderer.renderer_for_scene(sc, (1920, 1080), frames_in_flight=2); w.upload_state(); w.render_frame(); w.close()
print("after an unsharded 2-slot context: %.1f" % run())
