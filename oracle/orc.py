"""ctypes binding of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE -- PARITY UNPINNED (see art_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
# oracle/orc64.py executes this file again with _F64 = True: the same wrapper over liborc_f64.so, the build of art_oracle.c in which every
# float is a double (arrays, records, arithmetic); packing / presentation are not part of that build
_F64 = bool(globals().get("_F64", False))
REAL = np.float64 if _F64 else np.float32
CREAL = C.c_double if _F64 else C.c_float
_SO = "liborc_f64.so" if _F64 else os.environ.get("ORC_SO", "liborc.so")   # ORC_SO=liborc_asan.so: the sanitizer build (tests/test_oracle.py)


class OrcLight(C.Structure):
    _fields_ = [("pos", CREAL * 3), ("type", C.c_uint32), ("dir", CREAL * 3), ("casts_shadows", C.c_uint32),
                ("color", CREAL * 3), ("falloff_distance", CREAL), ("area_pos2", CREAL * 3), ("penumbra_angle", CREAL),
                ("area_pos3", CREAL * 3), ("umbra_angle", CREAL)]


class OrcCamera(C.Structure):
    _pack_ = 1
    _fields_ = [("view", CREAL * 16), ("view_inv", CREAL * 16), ("proj", CREAL * 16), ("proj_inv", CREAL * 16),
                ("camera_pos", CREAL * 3)]


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("primary_rays", "shadow_rays", "hit_pixels", "n_int_primary", "n_tri_primary",
                                          "n_int_shadow", "n_tri_shadow", "nonfinite_pixels")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


assert (C.sizeof(OrcLight), C.sizeof(OrcCamera)) == ((160, 536) if _F64 else (80, 268))


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "asan" if _SO == "liborc_asan.so" else _SO])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, _SO)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_add_primitive.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                              C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_scene_build.argtypes = [C.c_void_p, C.c_int]
        L.orc_scene_num_tris.argtypes = [C.c_void_p]
        L.orc_scene_num_tris.restype = C.c_uint32
        L.orc_scene_get_lbvh.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        L.orc_camera_from_params.argtypes = [C.c_void_p, C.c_void_p, CREAL, CREAL, CREAL, CREAL, C.c_void_p]
        L.orc_light_point.argtypes = [C.c_void_p, C.c_void_p, CREAL, C.c_int, C.c_void_p]
        L.orc_light_spot.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, CREAL, CREAL, CREAL, C.c_int, C.c_void_p]
        L.orc_light_directional.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_light_area.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, CREAL, CREAL, CREAL, C.c_int, C.c_void_p]
        L.orc_gen_primary.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_trace_any.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_packet_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_render_ao.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, CREAL, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        if not _F64:
            L.orc_pack_b10g11r11.argtypes = [C.c_void_p]; L.orc_pack_b10g11r11.restype = C.c_uint32
            L.orc_unpack_b10g11r11.argtypes = [C.c_uint32, C.c_void_p]
            L.orc_pack_f16.argtypes = [CREAL]; L.orc_pack_f16.restype = C.c_uint16
            L.orc_lpm_control_block.argtypes = [C.c_int, CREAL, CREAL, CREAL, CREAL, CREAL, C.c_void_p, C.c_void_p, C.c_void_p]
            L.orc_present.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_brdf_terms.argtypes = [CREAL] * 7 + [C.c_void_p]
        L.orc_light_eval.argtypes = [C.c_void_p] * 4
        _LIB = L
    return _LIB


def _f3(v):
    return (CREAL * 3)(*[float(x) for x in v])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def camera_from_params(pos, dir, aspect, fovy, znear, zfar) -> OrcCamera:
    cam = OrcCamera()
    lib().orc_camera_from_params(_f3(pos), _f3(dir), aspect, fovy, znear, zfar, C.byref(cam))
    return cam


def make_light(d: dict) -> OrcLight:
    L = lib()
    o = OrcLight()
    k = d["kind"]
    cs = int(bool(d.get("casts_shadows", False)))
    if k == "point":
        L.orc_light_point(_f3(d["pos"]), _f3(d["color"]), d["falloff"], cs, C.byref(o))
    elif k == "spot":
        L.orc_light_spot(_f3(d["pos"]), _f3(d["dir"]), _f3(d["color"]), d["falloff"], d["penumbra"], d["umbra"], cs, C.byref(o))
    elif k == "directional":
        L.orc_light_directional(_f3(d["dir"]), _f3(d["color"]), cs, C.byref(o))
    elif k == "area":
        L.orc_light_area(_f3(d["pos"]), _f3(d["pos2"]), _f3(d["pos3"]), int(bool(d.get("invert_normal", False))), _f3(d["color"]),
                         d["falloff"], d["penumbra"], d["umbra"], cs, C.byref(o))
    else:
        raise ValueError(k)
    return o


def make_lights(ds):
    arr = (OrcLight * max(1, len(ds)))()
    for i, d in enumerate(ds):
        arr[i] = make_light(d)
    return arr


class Scene:
    def __init__(self, primitives=None, morton_bits=30):
        self._L = lib()
        self.h = C.c_void_p(self._L.orc_scene_create())
        if primitives is not None:
            for p in primitives:
                self.add_primitive(p.verts, p.indices, p.tex, p.model)
            self.build(morton_bits)

    def __del__(self):
        if getattr(self, "h", None):
            self._L.orc_scene_destroy(self.h)
            self.h = None

    def add_primitive(self, verts, indices, tex, model):
        verts = np.ascontiguousarray(verts, dtype=REAL)
        indices = np.ascontiguousarray(indices)
        assert indices.dtype in (np.uint16, np.uint32)
        tex = np.ascontiguousarray(tex, dtype=np.uint8)
        model = np.ascontiguousarray(model, dtype=REAL)
        r = self._L.orc_scene_add_primitive(self.h, _ptr(verts), verts.shape[0], _ptr(indices), indices.size, indices.dtype.itemsize,
                                            _ptr(tex), tex.shape[2], tex.shape[1], _ptr(model))
        if r < 0:
            raise ValueError(f"orc_scene_add_primitive failed: {r}")
        return r

    def build(self, morton_bits=30):
        r = self._L.orc_scene_build(self.h, morton_bits)
        if r != 0:
            raise ValueError(f"orc_scene_build failed: {r}")

    @property
    def n_tris(self):
        return int(self._L.orc_scene_num_tris(self.h))

    def lbvh(self):
        T = self.n_tris
        NI = max(T - 1, 0)
        out = dict(leaf_gid=np.zeros(T, np.uint32), keys=np.zeros(T, np.uint64), child=np.zeros((NI, 2), np.int32),
                   node_lo=np.zeros((NI, 3), REAL), node_hi=np.zeros((NI, 3), REAL), leaf_lo=np.zeros((T, 3), REAL),
                   leaf_hi=np.zeros((T, 3), REAL), tri_verts=np.zeros((T, 9), REAL))
        self._L.orc_scene_get_lbvh(self.h, *[_ptr(out[k]) for k in ("leaf_gid", "keys", "child", "node_lo", "node_hi", "leaf_lo", "leaf_hi", "tri_verts")])
        return out

    def trace_closest(self, rays, mode=0):
        rays = np.ascontiguousarray(rays, dtype=REAL).reshape(-1, 8)
        n = rays.shape[0]
        tuv = np.zeros((n, 4), REAL)
        ids = np.zeros((n, 2), np.int32)
        ni, nt = C.c_uint64(), C.c_uint64()
        self._L.orc_trace_closest(self.h, _ptr(rays), n, mode, _ptr(tuv), _ptr(ids), C.byref(ni), C.byref(nt))
        return tuv, ids, int(ni.value), int(nt.value)

    def trace_any(self, rays, mode=0):
        rays = np.ascontiguousarray(rays, dtype=REAL).reshape(-1, 8)
        n = rays.shape[0]
        hit = np.zeros(n, np.uint8)
        ni, nt = C.c_uint64(), C.c_uint64()
        self._L.orc_trace_any(self.h, _ptr(rays), n, mode, _ptr(hit), C.byref(ni), C.byref(nt))
        return hit, int(ni.value), int(nt.value)

    def render(self, cam: OrcCamera, lights, n_lights, w, h, y0=0, y1=None, threads=1, debug=False, reuse=False):
        y1 = h if y1 is None else y1
        if reuse and getattr(self, "_bufs", None) is not None and self._bufs[0].shape == (h, w, 4):
            color, depth, normal = self._bufs          # timing loops: do not page-fault 75 MB of fresh output per frame
        else:
            color = np.zeros((h, w, 4), REAL)
            depth = np.zeros((h, w), REAL)
            normal = np.zeros((h, w, 4), REAL)
            if reuse:
                self._bufs = (color, depth, normal)
        tuv = np.zeros((h, w, 4), REAL) if debug else None
        ids = np.zeros((h, w, 2), np.int32) if debug else None
        sb = np.zeros((h, w), np.uint32) if debug else None
        st = OrcStats()
        self._L.orc_render(self.h, C.byref(cam), lights, n_lights, w, h, y0, y1, _ptr(color), _ptr(depth), _ptr(normal), _ptr(tuv), _ptr(ids),
                           _ptr(sb), C.byref(st), threads)
        out = dict(color=color, depth=depth, normal=normal, stats=st.as_dict())
        if debug:
            out.update(hit_tuv=tuv, hit_id=ids, shadow_bits=sb)
        return out


def packet_stats(scene: "Scene", cam: OrcCamera, lights, n_lights, w, h, block=(8, 8), threads=1):
    """visit counts per 8x8-pixel packet on the canonical LBVH (union of the packet's per-ray paths), plus the frame's per-ray stats"""
    out = (C.c_uint64 * 4)()
    st = OrcStats()
    lib().orc_packet_stats(scene.h, C.byref(cam), lights, n_lights, w, h, block[0], block[1], out, C.byref(st), threads)
    return dict(packet_nodes_primary=int(out[0]), packet_tris_primary=int(out[1]), packet_nodes_shadow=int(out[2]), packet_tris_shadow=int(out[3])), st.as_dict()


def render_ao(scene: "Scene", cam: OrcCamera, depth, normal, spp, radius, threads=1):
    h, w = depth.shape
    depth = np.ascontiguousarray(depth, REAL)
    normal = np.ascontiguousarray(normal, REAL)
    out = np.zeros((h, w), np.uint32)
    nr, ni, nt = C.c_uint64(), C.c_uint64(), C.c_uint64()
    lib().orc_render_ao(scene.h, C.byref(cam), w, h, _ptr(depth), _ptr(normal), spp, radius, _ptr(out), C.byref(nr), C.byref(ni), C.byref(nt), threads)
    return out, dict(ao_rays=int(nr.value), n_int_ao=int(ni.value), n_tri_ao=int(nt.value))


def present(color, ao=None):
    """-> (packed B10G11R11 colour [h,w] u32, BGRA8 [h,w,4])"""
    color = np.ascontiguousarray(color, REAL)
    h, w = color.shape[:2]
    packed = np.zeros((h, w), np.uint32)
    bgra = np.zeros((h, w, 4), np.uint8)
    aop = np.ascontiguousarray(ao, np.uint32) if ao is not None else None
    lib().orc_present(_ptr(color), _ptr(aop), w * h, _ptr(packed), _ptr(bgra))
    return packed, bgra


def pack_b10g11r11(rgb):
    a = np.ascontiguousarray(rgb, REAL)
    return int(lib().orc_pack_b10g11r11(_ptr(a)))


def unpack_b10g11r11(v):
    out = np.zeros(3, REAL)
    lib().orc_unpack_b10g11r11(int(v), _ptr(out))
    return out


def pack_f16(f):
    return int(lib().orc_pack_f16(float(f)))


def lpm_control_block(shoulder, soft_gap, hdr_max, exposure, contrast, shoulder_contrast, saturation, crosstalk):
    ctl = np.zeros(96, np.uint32)
    lib().orc_lpm_control_block(int(shoulder), soft_gap, hdr_max, exposure, contrast, shoulder_contrast, _f3(saturation), _f3(crosstalk), _ptr(ctl))
    return ctl


def gen_primary(cam: OrcCamera, w, h):
    rays = np.zeros((h * w, 8), REAL)
    lib().orc_gen_primary(C.byref(cam), w, h, _ptr(rays))
    return rays


def brdf_terms(NdotL, NdotV, NdotH, LdotH, nc_NdotV, nc_NdotL, alpha):
    out = np.zeros(4, REAL)
    lib().orc_brdf_terms(NdotL, NdotV, NdotH, LdotH, nc_NdotV, nc_NdotL, alpha, _ptr(out))
    return out


def light_eval(light: OrcLight, p):
    nn = np.zeros(3, REAL)
    rad = np.zeros(3, REAL)
    pp = np.asarray(p, REAL)
    lib().orc_light_eval(C.byref(light), _ptr(pp), _ptr(nn), _ptr(rad))
    return nn, rad
