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


def seam_scene(scenes):
    """one quad whose uv run from -1.25 to 2.25 (and -0.75 to 1.75) under a 4 x 5 texture of unrelated texels: nearly every pixel's bilinear footprint sits on a texel boundary,
    a fifth of them on the REPEAT wrap (tests/test_oracle.py: against numpy in fp64; tests/test_gpu_parity.py: the GPU against the oracle)"""
    import math
    rng = np.random.default_rng(7)
    tw, th = 4, 5
    tex = np.zeros((3, th, tw, 4), np.uint8)
    tex[0] = rng.integers(20, 256, (th, tw, 4))                                     # albedo: unrelated texels
    tex[1, ..., 0] = 255; tex[1, ..., 1] = rng.integers(80, 230, (th, tw)); tex[1, ..., 2] = rng.integers(0, 2, (th, tw)) * 255   # occlusion, roughness, metallic
    nm = rng.normal(0, 0.35, (th, tw, 3)); nm[..., 2] = 1.0; nm /= np.linalg.norm(nm, axis=-1, keepdims=True)
    tex[2, ..., :3] = np.round((nm * 0.5 + 0.5) * 255); tex[2, ..., 3] = 255
    mb = scenes.MeshBuilder()
    mb.add([(-0.8, -0.6, 0.5), (0.8, -0.6, 0.5), (0.8, 0.6, 0.7), (-0.8, 0.6, 0.7)], [(-1.25, -0.75), (2.25, -0.75), (2.25, 1.75), (-1.25, 1.75)],
           [(0, 0, -1)] * 4, [(1, 0, 0, 1)] * 4, [0, 1, 2, 0, 2, 3])
    prim = mb.finish(tex)
    return scenes.Scene("seams", [prim], dict(pos=(0.0, 0.0, -0.6), dir=(0.0, 0.0, 1.0), fovy=math.pi / 2, znear=0.1, zfar=1000.0),
                        [dict(kind="point", pos=(0.3, -0.2, -0.3), color=(6.0, 5.0, 4.0), falloff=4.0, casts_shadows=False)])
