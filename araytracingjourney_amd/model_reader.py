"""ctypes mirror of the reference's ModelReader / GltfModelReader (model_reader/*.rs) over libart's art_glb_* entry points."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import ArtGlbCopyInfo

VERTICES, TEX_COORDS, NORMALS, TANGENTS, INDICES = 1, 2, 4, 8, 16   # MeshAttributeType (model_reader.rs:6-12)
ALBEDO, ORM, NORMAL, EMISSIVE = 1, 2, 4, 8                           # TextureType (model_reader.rs:14-19)
COERCE_NONE, COERCE_R8G8B8A8, COERCE_B8G8R8A8, COERCE_B8G8R8 = 0, 1, 2, 3


def _check(code):
    if code != 0:
        raise _lib.ArtError(code, _lib.load().art_glb_last_error().decode("utf-8", "replace"))


class GltfModelReader:
    def __init__(self, file_path, normalize_vectors=True, coerce_image_to_format=COERCE_B8G8R8A8):  # open (gltf_model_reader.rs:55)
        self._L = _lib.load()
        self._h = C.c_void_p()
        _check(self._L.art_glb_open(str(file_path).encode(), int(normalize_vectors), int(coerce_image_to_format), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.art_glb_close(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def primitive_count(self):
        n = C.c_uint32()
        _check(self._L.art_glb_primitive_count(self._h, C.byref(n)))
        return n.value

    def copy_model_data_to_ptr(self, mesh_attributes, textures, copy=True):  # gltf_model_reader.rs:156-281
        n = self.primitive_count()
        infos = (ArtGlbCopyInfo * max(1, n))()
        total = C.c_size_t()
        _check(self._L.art_glb_copy_model_data(self._h, mesh_attributes, textures, None, 0, infos, n, C.byref(total)))
        data = None
        if copy:
            data = np.zeros(total.value, np.uint8)
            _check(self._L.art_glb_copy_model_data(self._h, mesh_attributes, textures, data.ctypes.data_as(C.c_void_p), data.size, infos, n, C.byref(total)))
        return data, [infos[i] for i in range(n)]

    def get_primitives_bounding_sphere(self):  # gltf_model_reader.rs:283-399
        c = (C.c_float * 3)()
        r = C.c_float()
        _check(self._L.art_glb_bounding_sphere(self._h, c, C.byref(r)))
        return np.array(list(c), np.float32), float(r.value)


def permute_pixels(src, src_texel_size, src_to_dst_map: dict, dst_texel_size):  # gltf_model_reader.rs:542-573
    src = np.ascontiguousarray(src, np.uint8)
    m = np.full(max(src_to_dst_map) + 1, -1, np.int32)
    for k, v in src_to_dst_map.items():
        m[k] = v
    out = np.zeros((src.size // src_texel_size) * dst_texel_size, np.uint8)
    _check(_lib.load().art_glb_permute_pixels(src.ctypes.data_as(C.c_void_p), src.size, src_texel_size, m.ctypes.data_as(C.c_void_p), m.size, dst_texel_size,
                                              out.ctypes.data_as(C.c_void_p), out.size))
    return out
