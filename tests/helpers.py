import numpy as np


def oracle_for(orc, scene, n_lights=None, morton_bits=30):
    S = orc.Scene(scene.primitives, morton_bits=morton_bits)
    ls = scene.lights if n_lights is None else scene.lights[:n_lights]
    return S, orc.make_lights(ls), len(ls)


def oracle_camera(orc, scene, w, h):
    c = scene.camera
    return orc.camera_from_params(c["pos"], c["dir"], w / h, c["fovy"], c["znear"], c["zfar"])


def random_rays(n, seed, radius=2.5):
    """Deterministic rays from points on a sphere around the scene towards points near the origin."""
    k = np.arange(n, dtype=np.float64)

    def h(a):
        x = np.sin(k * a + seed * 0.618) * 43758.5453
        return x - np.floor(x)
    th, ph = h(12.9898) * 2 * np.pi, np.arccos(2 * h(78.233) - 1)
    o = radius * np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], 1)
    tgt = (np.stack([h(3.1), h(5.7), h(9.3)], 1) - 0.5) * 1.5
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = o
    rays[:, 3] = 0.001
    rays[:, 4:7] = d
    rays[:, 7] = 100.0
    # a third of the rays start inside the scene
    inside = (np.arange(n) % 3) == 0
    rays[inside, 0:3] = (tgt[inside] * 0.6).astype(np.float32)
    return rays


def device_to_host(ptr, nbytes):
    """bytes at a raw device pointer (art_device_*) -> numpy uint8; the process's one HIP runtime (loaded by torch / libart)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = np.empty(nbytes, np.uint8)
    rc = hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), nbytes, 2)   # hipMemcpyDeviceToHost
    assert rc == 0, f"hipMemcpy failed: {rc}"
    return out
