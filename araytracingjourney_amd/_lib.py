"""ctypes loader of libart.so (the C ABI of include/art.h).  Fails loudly: there is no Python or CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ART_LIB_PATH") or os.path.join(_HERE, "libart.so")   # ART_LIB_PATH: another build of the same sources, for A/B measurements (tools/); libart itself reads no environment

ART_OK, ART_E_INVALID, ART_E_STATE, ART_E_NO_DEVICE, ART_E_HIP, ART_E_NOMEM = 0, -1, -2, -3, -4, -5
ART_FLAG_KEEP_DEBUG = 1
ART_FLAG_PACKED_TILES = 4  # sharded: the gather payload is the B10G11R11 colour (4 B per pixel)
ART_FLAG_FIXED_WAVES = 16  # one wave per 8x8 block always (default: the adaptive wave plan of the fused frame)
ART_FLAG_TILE_OUTPUT = 32  # compact tile buffer even for an unsharded frame (a one-rank art_mgpu job)
ART_FLAG_DYNAMIC_SCENE = 64  # models will move / leave / re-enter: art_scene_build also makes the ring of structure versions (else the first moved frame does)
ART_FLAG_FAST_BUILD = 2  # keep the LBVH topology in the traversal nodes (default: binned-SAH rebuild, PREFER_FAST_TRACE)


class ArtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libart error {code}: {msg}")
        self.code = code


class ArtVertex(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("uv", C.c_float * 2), ("normal", C.c_float * 3), ("tangent", C.c_float * 4)]


class ArtLight(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("type", C.c_uint32), ("dir", C.c_float * 3), ("casts_shadows", C.c_uint32),
                ("color", C.c_float * 3), ("falloff_distance", C.c_float), ("area_pos2", C.c_float * 3), ("penumbra_angle", C.c_float),
                ("area_pos3", C.c_float * 3), ("umbra_angle", C.c_float)]


class ArtCamera(C.Structure):
    _pack_ = 1
    _fields_ = [("view", C.c_float * 16), ("view_inv", C.c_float * 16), ("proj", C.c_float * 16), ("proj_inv", C.c_float * 16),
                ("camera_pos", C.c_float * 3)]


class ArtConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32), ("morton_bits", C.c_uint32),
                ("shard_rank", C.c_uint32), ("shard_count", C.c_uint32), ("flags", C.c_uint32), ("frames_in_flight", C.c_uint32), ("root_relief", C.c_uint32)]


class ArtStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("hit_pixels", C.c_uint64), ("ao_rays", C.c_uint64),
                ("num_triangles", C.c_uint32), ("num_primitives", C.c_uint32), ("num_nodes", C.c_uint32), ("frame_launches", C.c_uint32),
                ("build_ms", C.c_float), ("frame_ms", C.c_float), ("trace_primary_ms", C.c_float), ("shade_ms", C.c_float),
                ("trace_shadow_ms", C.c_float), ("accumulate_ms", C.c_float), ("ao_ms", C.c_float), ("split_blocks", C.c_uint32),
                ("refit_ms", C.c_float), ("refit_cost_ratio", C.c_float), ("refits", C.c_uint32), ("rebuilds", C.c_uint32), ("first_move_ms", C.c_float), ("versions_ms", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if not n.startswith("reserved")}


class ArtGlbCopyInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("mesh_buffer_offset", "mesh_size", "indices_buffer_offset", "indices_size", "image_buffer_offset", "image_size")] + \
               [(n, C.c_uint32) for n in ("single_mesh_element_size", "single_index_size", "image_format", "image_width", "image_height", "image_mip_levels",
                                          "image_layers", "reserved")]


class ArtTuning(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("frame_form", "tree_builder", "packet_wide", "primary_walk", "shadow_walk", "ao_walk", "block_order", "fixed_waves",
                                          "split_fixed_steps", "split_min_steps")] + [("split_alpha", C.c_float)] + \
               [(n, C.c_uint32) for n in ("ao_entry_off", "trace_chunk", "trace_refill", "trace_blocks", "hw_queues", "log", "wide_builder", "as_versions")] + [("refit_rebuild_ratio", C.c_float), ("trace_leaf_batch", C.c_uint32), ("plan_moving_interval", C.c_uint32), ("refit_streams", C.c_uint32), ("refit_fold_nodes", C.c_uint32)]


class ArtLayout(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("width", "height", "frames_in_flight", "frames_per_launch", "shard_rank", "shard_count", "tiles_owned", "tiles_padded",
                                          "tile_bytes", "reserved")]


# int32_t (*)(void *user, const void *send_dev, size_t bytes, void *recv_dev, void *hip_stream)
ArtMgpuExchangeFn = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p)
ART_MGPU_ID_BYTES = 128
ART_MGPU_SHARED, ART_MGPU_DEDICATED = 0, 1
ART_MGPU_RCCL, ART_MGPU_HOST_EXCHANGE = 0, 1
ART_MGPU_ROOT_RANK0, ART_MGPU_ROOT_SPREAD = 0, 1


class ArtMgpuConfig(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("compositor", C.c_uint32), ("launches_per_gather", C.c_uint32), ("tile_buffers", C.c_uint32),
                ("transport", C.c_uint32), ("exchange", ArtMgpuExchangeFn), ("exchange_user", C.c_void_p), ("roots", C.c_uint32), ("reserved", C.c_uint32)]


assert C.sizeof(ArtVertex) == 48 and C.sizeof(ArtLight) == 80 and C.sizeof(ArtCamera) == 268

# every symbol include/art.h declares -- the boundary: (name, restype, argtypes)
_P, _U32, _I32, _F, _SZ = C.c_void_p, C.c_uint32, C.c_int32, C.c_float, C.c_size_t
SYMBOLS = {
    "art_last_error": (C.c_char_p, []),
    "art_device_count": (_I32, []),
    "art_create": (_I32, [_P, _P]),
    "art_destroy": (_I32, [_P]),
    "art_set_stream": (_I32, [_P, _P]),
    "art_scene_add_primitive": (_I32, [_P, _P, _U32, _P, _U32, _U32, _P, _U32, _U32, _P, _P]),
    "art_scene_clear": (_I32, [_P]),
    "art_scene_set_primitive_enabled": (_I32, [_P, _U32, _I32]),
    "art_scene_needs_build": (_I32, [_P]),
    "art_scene_set_model_matrix": (_I32, [_P, _U32, _U32, _P]),
    "art_scene_build": (_I32, [_P]),
    "art_set_camera": (_I32, [_P, _P]),
    "art_camera_from_params": (_I32, [_P, _P, _F, _F, _F, _F, _P]),
    "art_set_lights": (_I32, [_P, _P, _U32]),
    "art_light_point": (_I32, [_P, _P, _F, _I32, _P]),
    "art_light_spot": (_I32, [_P, _P, _P, _F, _F, _F, _I32, _P]),
    "art_light_directional": (_I32, [_P, _P, _I32, _P]),
    "art_light_area": (_I32, [_P, _P, _P, _I32, _P, _F, _F, _F, _I32, _P]),
    "art_resize": (_I32, [_P, _U32, _U32]),
    "art_trace": (_I32, [_P]),
    "art_sync": (_I32, [_P]),
    "art_present": (_I32, [_P]),
    "art_read_present": (_I32, [_P, _P, _SZ]),
    "art_read_packed": (_I32, [_P, _P, _P, _P]),
    "art_lpm_control_block": (_I32, [_I32, _F, _F, _F, _F, _F, _P, _P, _P]),
    "art_trace_ao": (_I32, [_P, _U32, _F]),
    "art_read_ao": (_I32, [_P, _P, _SZ]),
    "art_read_color": (_I32, [_P, _P, _SZ]),
    "art_read_depth": (_I32, [_P, _P, _SZ]),
    "art_read_normal": (_I32, [_P, _P, _SZ]),
    "art_device_color": (_I32, [_P, _P, _P]),
    "art_device_depth": (_I32, [_P, _P, _P]),
    "art_device_normal": (_I32, [_P, _P, _P]),
    "art_shard_layout": (_I32, [_U32, _U32, _U32, _U32, _U32, _P, _U32, _P, _P]),
    "art_shard_tile_count": (_I32, [_P, _P, _P]),
    "art_device_color_tiles": (_I32, [_P, _P, _P]),
    "art_set_frames_per_launch": (_I32, [_P, _U32]),
    "art_set_camera_batch": (_I32, [_P, _P, _U32]),
    "art_set_read_frame": (_I32, [_P, _U32]),
    "art_frames_in_flight": (_I32, [_P, _P, _P]),
    "art_frames_done": (_I32, [_P, C.c_uint64, _U32, _P, _P]),
    "art_untile_gathered": (_I32, [_P, _P, _U32, _P, _P]),
    "art_get_stats": (_I32, [_P, _P]),
    "art_get_layout": (_I32, [_P, _P]),
    "art_mgpu_shard": (_I32, [_U32, _U32, _U32, _P, _P]),
    "art_mgpu_unique_id": (_I32, [_P]),
    "art_mgpu_create": (_I32, [_P, _P, _P, _P]),
    "art_mgpu_trace": (_I32, [_P]),
    "art_mgpu_flush": (_I32, [_P]),
    "art_mgpu_device_frame": (_I32, [_P, _P, _P]),
    "art_mgpu_read_frame": (_I32, [_P, _P, _SZ]),
    "art_mgpu_counts": (_I32, [_P, _P, _P, _P]),
    "art_mgpu_destroy": (_I32, [_P]),
    "art_glb_last_error": (C.c_char_p, []),
    "art_glb_open": (_I32, [C.c_char_p, _I32, _I32, _P]),
    "art_glb_close": (_I32, [_P]),
    "art_glb_primitive_count": (_I32, [_P, _P]),
    "art_glb_copy_model_data": (_I32, [_P, _U32, _U32, _P, _SZ, _P, _U32, _P]),
    "art_glb_bounding_sphere": (_I32, [_P, _P, _P]),
    "art_glb_permute_pixels": (_I32, [_P, _SZ, _U32, _P, _U32, _U32, _P, _SZ]),
    "art_scene_add_glb": (_I32, [_P, _P, _P, _P, _P]),
}
# include/art_parity.h: the parity / rehearsal / measurement surface (same library, not part of the boundary)
PARITY_SYMBOLS = {
    "art_bind_color_tiles": (_I32, [_P, _U32, _P, _SZ]),
    "art_bind_color_tiles_pair": (_I32, [_P, _U32, _P, _P, _SZ]),
    "art_bind_color_tiles_ring": (_I32, [_P, _U32, _P, _U32, _SZ]),
    "art_set_graph_mode": (_I32, [_P, _I32]),
    "art_stream_wait_frame": (_I32, [_P, _P]),
    "art_wait_external_event": (_I32, [_P, _P]),
    "art_trace_for_stream": (_I32, [_P, _P, _P]),
    "art_collect_timings": (_I32, [_P, _P, _P]),
    "art_sample_wave_steps": (_I32, [_P, _P, _P, _U32, _P]),
    "art_read_color_tiles": (_I32, [_P, _P, _SZ]),
    "art_untile_gathered_strided": (_I32, [_P, _P, _U32, _U32, _P, _P]),
    "art_untile_gathered_frames": (_I32, [_P, _P, _U32, _U32, _U32, _P, _P]),
    "art_set_tuning": (_I32, [_P, _P]),
    "art_timestamp_mark": (_I32, [_P, _U32]),
    "art_timestamp_elapsed": (_I32, [_P, _P]),
    "art_read_hits": (_I32, [_P, _P, _P, _SZ]),
    "art_read_shadow_bits": (_I32, [_P, _P, _SZ]),
    "art_query_closest": (_I32, [_P, _P, _U32, _P, _P]),
    "art_query_any": (_I32, [_P, _P, _U32, _P]),
    "art_get_lbvh": (_I32, [_P] + [_P] * 7),
    "art_get_traversal_tree": (_I32, [_P, _P, _P, _P]),
    "art_get_wide_nodes": (_I32, [_P, _P, _P, _SZ, _P]),
    "art_mgpu_pending": (_I32, [_P, _P, _P]),
}

_lib = None


def load():
    """Load libart.so; raises if the HIP extension has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C araytracingjourney_amd/csrc` (libart has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SYMBOLS.items()) + list(PARITY_SYMBOLS.items()):
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != ART_OK:
        raise ArtError(code, load().art_last_error().decode("utf-8", "replace"))
