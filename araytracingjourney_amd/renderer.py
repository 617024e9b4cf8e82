"""Host-side mirror of the reference's renderer API over the libart C ABI.

Names follow /root/reference/src/vk_renderer: `Renderer` ~ `VulkanTempleRayTracedRenderer` (renderer.rs:121-137:
new / add_model / prepare_first_frame / render_frame / camera_mut / lights_mut), `Camera` ~ `VkCamera`
(vk_camera.rs:128-193), `Lights`, `PointLight`, `SpotLight`, `DirectionalLight`, `AreaLight` ~ lights.rs.
All arithmetic (matrices, light records, tracing) happens inside libart; this module only marshals.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import _lib
from ._lib import ArtCamera, ArtConfig, ArtLight, ArtStats, check


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ------------------------------------------------------------------------------------------------ lights (lights.rs)
class PointLight:
    def __init__(self, pos, color, falloff_distance, casts_shadows):  # lights.rs:103
        self.pos, self.color, self.falloff_distance, self.casts_shadows = tuple(pos), tuple(color), float(falloff_distance), bool(casts_shadows)

    def get_light_shader_data(self) -> ArtLight:  # lights.rs:144-159
        o = ArtLight()
        check(_lib.load().art_light_point(_f3(self.pos), _f3(self.color), self.falloff_distance, int(self.casts_shadows), C.byref(o)))
        return o


class SpotLight:
    def __init__(self, pos, dir, color, falloff_distance, penumbra_umbra_angles, casts_shadows):  # lights.rs:171
        self.pos, self.dir, self.color = tuple(pos), tuple(dir), tuple(color)
        self.falloff_distance, self.penumbra_umbra_angles, self.casts_shadows = float(falloff_distance), tuple(penumbra_umbra_angles), bool(casts_shadows)

    def get_light_shader_data(self) -> ArtLight:  # lights.rs:228-243
        o = ArtLight()
        check(_lib.load().art_light_spot(_f3(self.pos), _f3(self.dir), _f3(self.color), self.falloff_distance, self.penumbra_umbra_angles[0],
                                         self.penumbra_umbra_angles[1], int(self.casts_shadows), C.byref(o)))
        return o


class DirectionalLight:
    def __init__(self, dir, color, casts_shadows):  # lights.rs:252
        self.dir, self.color, self.casts_shadows = tuple(dir), tuple(color), bool(casts_shadows)

    def get_light_shader_data(self) -> ArtLight:  # lights.rs:281-296
        o = ArtLight()
        check(_lib.load().art_light_directional(_f3(self.dir), _f3(self.color), int(self.casts_shadows), C.byref(o)))
        return o


class AreaLight:
    def __init__(self, pos, pos2, pos3, invert_normal, color, falloff_distance, penumbra_umbra_angles, casts_shadows):  # lights.rs:310
        self.pos, self.pos2, self.pos3, self.invert_normal, self.color = tuple(pos), tuple(pos2), tuple(pos3), bool(invert_normal), tuple(color)
        self.falloff_distance, self.penumbra_umbra_angles, self.casts_shadows = float(falloff_distance), tuple(penumbra_umbra_angles), bool(casts_shadows)

    def get_light_shader_data(self) -> ArtLight:  # lights.rs:383-403
        o = ArtLight()
        check(_lib.load().art_light_area(_f3(self.pos), _f3(self.pos2), _f3(self.pos3), int(self.invert_normal), _f3(self.color), self.falloff_distance,
                                         self.penumbra_umbra_angles[0], self.penumbra_umbra_angles[1], int(self.casts_shadows), C.byref(o)))
        return o


class Lights:
    """lights.rs:4-67.  Serialisation order point, spot, directional, area; unlike the reference's
    copy_lights_shader_data (lights.rs:24-47, which writes every light of a kind into one slot) each light gets
    its own slot -- identical whenever there is at most one light per kind."""

    def __init__(self):
        self.point_lights, self.spot_lights, self.directional_lights, self.area_lights = [], [], [], []

    def get_point_lights_mut(self):
        return self.point_lights

    def get_spot_lights_mut(self):
        return self.spot_lights

    def get_directional_lights_mut(self):
        return self.directional_lights

    def get_area_lights_mut(self):
        return self.area_lights

    def get_lights_count(self):
        return len(self.point_lights) + len(self.spot_lights) + len(self.directional_lights) + len(self.area_lights)

    def copy_lights_shader_data(self):
        all_ = self.point_lights + self.spot_lights + self.directional_lights + self.area_lights
        arr = (ArtLight * max(1, len(all_)))()
        for i, l in enumerate(all_):
            arr[i] = l.get_light_shader_data()
        return arr, len(all_)

    def push_dict(self, d):
        k = d["kind"]
        if k == "point":
            self.point_lights.append(PointLight(d["pos"], d["color"], d["falloff"], d["casts_shadows"]))
        elif k == "spot":
            self.spot_lights.append(SpotLight(d["pos"], d["dir"], d["color"], d["falloff"], (d["penumbra"], d["umbra"]), d["casts_shadows"]))
        elif k == "directional":
            self.directional_lights.append(DirectionalLight(d["dir"], d["color"], d["casts_shadows"]))
        elif k == "area":
            self.area_lights.append(AreaLight(d["pos"], d["pos2"], d["pos3"], d.get("invert_normal", False), d["color"], d["falloff"],
                                              (d["penumbra"], d["umbra"]), d["casts_shadows"]))
        else:
            raise ValueError(k)


# ------------------------------------------------------------------------------------------------ camera (vk_camera.rs)
class Camera:
    def __init__(self, pos, dir, aspect, fovy, znear, zfar):  # VkCamera::new, defaults renderer.rs:222-231
        self._pos, self._dir, self._aspect, self._fovy, self._znear, self._zfar = tuple(pos), tuple(dir), aspect, fovy, znear, zfar
        self.needs_update = True
        self._block = ArtCamera()

    def set_pos(self, pos):
        self._pos, self.needs_update = tuple(pos), True

    def set_dir(self, dir):
        self._dir, self.needs_update = tuple(dir), True  # normalised inside libart like vk_camera.rs:133-136

    def set_aspect(self, aspect):
        self._aspect, self.needs_update = aspect, True

    def set_fovy(self, fovy):
        self._fovy, self.needs_update = fovy, True

    def set_znear(self, znear):
        self._znear, self.needs_update = znear, True

    def set_zfar(self, zfar):
        self._zfar, self.needs_update = zfar, True

    def pos(self):
        return self._pos

    def dir(self):
        return self._dir

    def aspect(self):
        return self._aspect

    def fovy(self):
        return self._fovy

    def update_host_buffer(self) -> ArtCamera:  # vk_camera.rs:104-126
        if self.needs_update:
            check(_lib.load().art_camera_from_params(_f3(self._pos), _f3(self._dir), self._aspect, self._fovy, self._znear, self._zfar, C.byref(self._block)))
            self.needs_update = False
        return self._block

    def view_matrix(self):  # column-major 4x4 as numpy [4,4] (row, col)
        return np.array(self.update_host_buffer().view, dtype=np.float32).reshape(4, 4).T

    def perspective_matrix(self):
        return np.array(self.update_host_buffer().proj, dtype=np.float32).reshape(4, 4).T


# ------------------------------------------------------------------------------------------------ renderer (renderer.rs)
class Sphere:
    """model_reader.rs:100-146"""

    def __init__(self, center, radius):
        self.center, self.radius = np.asarray(center, np.float32), float(radius)

    def get_distance_from_point(self, point):  # model_reader.rs:124-126
        return float(np.linalg.norm(self.center - np.asarray(point, np.float32))) - self.radius

    def transform(self, m):  # model_reader.rs:128-141; m: row-major 3x4
        m = np.asarray(m, np.float32).reshape(3, 4)
        scale = max(float(np.linalg.norm(m[:, k])) for k in range(3))
        return Sphere(m[:, :3] @ self.center + m[:, 3], scale * self.radius)


STORAGE, HOST, DEVICE = "Storage", "Host", "Device"


class Model:
    """VkModel's residency state machine (vk_model.rs:280-345, states :27-275): Storage <-> Host <-> Device by the distance between the
    camera and the model's bounding sphere.  Only Device models are instanced in the acceleration structure (renderer.rs:640-651)."""

    def __init__(self, primitive_ids, sphere, reload=None, renderer=None, model_matrix=None, object_sphere=None):
        self.primitive_ids = list(primitive_ids)
        self.model_bounding_sphere = sphere
        self._object_sphere = object_sphere     # the reader's sphere, before any model matrix (set_model_matrix transforms THIS one)
        self._renderer = renderer               # the libart context that instances the primitives
        self.model_matrix = None if model_matrix is None else np.array(model_matrix, np.float32).reshape(3, 4)
        self.state = HOST                       # VkModel::new goes Storage -> Host (vk_model.rs:324-329)
        self.needs_cb_submit = False            # a transition to or from Device changes what the next build must contain
        self._reload = reload                   # Storage -> Host: how to read the model again (GLB path), None for in-memory models
        self._instanced = True                  # libart instances a primitive from the moment it is added

    def update_model_status(self, camera_pos):  # vk_model.rs:334-345
        d = self.model_bounding_sphere.get_distance_from_point(camera_pos)
        want = DEVICE if d <= 10.0 else (HOST if d <= 20.0 else STORAGE)
        if (want == DEVICE) != (self.state == DEVICE):
            self.needs_cb_submit = True
        self.state = want

    def set_model_matrix(self, matrix):
        """VkModel::set_model_matrix (vk_model.rs:461-466): the row-major 3x4 object -> world matrix of the model's instance, and the bounding sphere
        that goes with it.  Fixed against the reference: it transforms the sphere it HOLDS -- already transformed by the previous matrix -- by the new one
        (:463-465), so a model that is moved every frame compounds its matrices (a scale of 2 doubles the radius per call); here the reader's
        object-space sphere is transformed, which is the same for the one call the reference's main.rs makes.  The reference rebuilds its TLAS every
        frame for a moved model (renderer.rs:637-651); libart refits its structure on the device in front of the next frame (art_scene_set_model_matrix)."""
        m = np.ascontiguousarray(matrix, dtype=np.float32).reshape(3, 4)
        self.model_matrix = m.copy()
        self.model_bounding_sphere = (self._object_sphere if self._object_sphere is not None else self.model_bounding_sphere).transform(m)
        if self._renderer is not None and self.primitive_ids:
            ids = sorted(self.primitive_ids)
            runs, start = [], ids[0]                # consecutive ids travel as one call
            for a, b in zip(ids, ids[1:] + [None]):
                if b is None or b != a + 1:
                    runs.append((start, a - start + 1)); start = b
            for first, n in runs:
                check(self._renderer._L.art_scene_set_model_matrix(self._renderer._ctx, first, n, _ptr(m)))

    def get_transform_model_matrix(self):           # vk_model.rs:358-363
        return None if self.model_matrix is None else self.model_matrix.copy()

    def needs_command_buffer_submission(self):
        return self.needs_cb_submit

    def reset_command_buffer_submission_status(self):
        self.needs_cb_submit = False


class Renderer:
    """VulkanTempleRayTracedRenderer (renderer.rs:121-137) on libart: same call order, no window/swapchain."""

    def __init__(self, extent, device=-1, shard=(0, 1), morton_bits=0, keep_debug=False, frames_in_flight=1, fast_build=False, packed_tiles=False, fixed_waves=False, tile_output=False, tuning=None, root_relief=0, dynamic_scene=False):
        self._L = _lib.load()
        w, h = extent
        cfg = ArtConfig(device=device, width=w, height=h, morton_bits=morton_bits, shard_rank=shard[0], shard_count=shard[1],
                        flags=(_lib.ART_FLAG_KEEP_DEBUG if keep_debug else 0) | (_lib.ART_FLAG_FAST_BUILD if fast_build else 0) | (_lib.ART_FLAG_PACKED_TILES if packed_tiles else 0) | (_lib.ART_FLAG_FIXED_WAVES if fixed_waves else 0) | (_lib.ART_FLAG_TILE_OUTPUT if tile_output else 0) | (_lib.ART_FLAG_DYNAMIC_SCENE if dynamic_scene else 0),
                        frames_in_flight=frames_in_flight, root_relief=root_relief)
        self._ctx = C.c_void_p()
        check(self._L.art_create(C.byref(cfg), C.byref(self._ctx)))
        # The host tells the library how many hardware queues it asked HIP for (libart itself reads no environment variable); `tuning` picks one of the
        # equivalent forms of the path (ArtTuning: staged / per-ray frames, host-built tree, wave-plan targets ...) -- tests and sweeps only.
        t = _lib.ArtTuning(**dict(tuning or {}))
        if not t.hw_queues:
            t.hw_queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)
        check(self._L.art_set_tuning(self._ctx, C.byref(t)))
        self.extent = (w, h)
        self.shard = shard
        self.packed_tiles = packed_tiles
        # defaults of renderer.rs:222-231
        self._camera = Camera((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), w / h, math.pi / 2, 0.1, 1000.0)
        self._lights = Lights()
        self._models = []

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.art_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # renderer.rs:346 -- the reference takes a .glb path; here a model is the list of primitives the GLB reader yields
    def add_model(self, primitives, model_matrix=None):
        ids = []
        for p in primitives:
            verts = np.ascontiguousarray(p.verts, dtype=np.float32)
            idx = np.ascontiguousarray(p.indices)
            if idx.dtype not in (np.uint16, np.uint32):
                raise TypeError("indices must be uint16 or uint32")
            tex = np.ascontiguousarray(p.tex, dtype=np.uint8)
            m = np.ascontiguousarray(model_matrix if model_matrix is not None else p.model, dtype=np.float32)
            pid = C.c_uint32()
            check(self._L.art_scene_add_primitive(self._ctx, _ptr(verts), verts.shape[0], _ptr(idx), idx.size, idx.dtype.itemsize, _ptr(tex),
                                                  tex.shape[2], tex.shape[1], _ptr(m), C.byref(pid)))
            ids.append(pid.value)
        lo = np.min([np.asarray(p.verts)[:, :3].min(0) for p in primitives], 0)
        hi = np.max([np.asarray(p.verts)[:, :3].max(0) for p in primitives], 0)
        c = 0.5 * (lo + hi)
        rad = max(float(np.linalg.norm(np.asarray(p.verts)[:, :3] - c, axis=1).max()) for p in primitives)
        mm = model_matrix if model_matrix is not None else primitives[0].model
        self._models.append(Model(ids, Sphere(c, rad).transform(mm), renderer=self, model_matrix=mm, object_sphere=Sphere(c, rad)))
        return ids

    def add_model_glb(self, reader, model_matrix):
        """renderer.rs:346 with a GltfModelReader (opened with normalize + B8G8R8A8 coercion like vk_model.rs:498-504)"""
        first, n = C.c_uint32(), C.c_uint32()
        m = np.ascontiguousarray(model_matrix, dtype=np.float32)
        r = self._L.art_scene_add_glb(self._ctx, reader._h, _ptr(m), C.byref(first), C.byref(n))
        if r != 0:
            raise _lib.ArtError(r, self._L.art_glb_last_error().decode("utf-8", "replace"))
        ids = list(range(first.value, first.value + n.value))
        c, rad = reader.get_primitives_bounding_sphere()   # vk_model.rs:501, then set_model_matrix (:461-466)
        self._models.append(Model(ids, Sphere(c, rad).transform(model_matrix), renderer=self, model_matrix=model_matrix, object_sphere=Sphere(c, rad)))
        return ids

    def models_mut(self):
        return self._models

    def camera_mut(self) -> Camera:
        return self._camera

    def lights_mut(self) -> Lights:
        return self._lights

    def prepare_first_frame(self):  # renderer.rs:356: uploads + BLAS/TLAS builds
        self.update_models_status(build=False)
        check(self._L.art_scene_build(self._ctx))

    def update_models_status(self, build=True):
        """renderer.rs:637-651: every model decides its residency from the camera position; the acceleration structure is rebuilt over the
        Device models when that set changed -- by libart's refit when the models concerned were part of the last build (no build: art_scene_set_primitive_enabled),
        by art_scene_build otherwise.  Returns True when it was built again."""
        changed = False
        for m in self._models:
            m.update_model_status(self._camera.pos())
            m.reset_command_buffer_submission_status()
            if (m.state == DEVICE) != m._instanced:      # what libart holds differs from the model's state
                m._instanced = m.state == DEVICE
                for pid in m.primitive_ids:
                    check(self._L.art_scene_set_primitive_enabled(self._ctx, pid, 1 if m._instanced else 0))
                changed = True
        rebuilt = False
        if changed and build and self.needs_build():   # (a model that was part of the last build leaves / re-enters by the next frame's refit: nothing to build)
            check(self._L.art_scene_build(self._ctx))
            rebuilt = True
        return rebuilt

    def needs_build(self) -> bool:
        nb = self._L.art_scene_needs_build(self._ctx)
        if nb < 0:
            check(nb)
        return bool(nb)

    def set_stream(self, hip_stream_ptr):
        check(self._L.art_set_stream(self._ctx, C.c_void_p(hip_stream_ptr)))

    def resize(self, extent):
        check(self._L.art_resize(self._ctx, extent[0], extent[1]))
        self.extent = tuple(extent)
        self._camera.set_aspect(extent[0] / extent[1])

    def upload_state(self):
        """camera.update_host_buffer + lights.update_host_and_device_buffer (renderer.rs:374, :677)."""
        check(self._L.art_set_camera(self._ctx, C.byref(self._camera.update_host_buffer())))
        arr, n = self._lights.copy_lights_shader_data()
        check(self._L.art_set_lights(self._ctx, arr, n))

    def trace(self):
        """record + submit of lightning_layer.trace_rays (renderer.rs:679-686); asynchronous."""
        check(self._L.art_trace(self._ctx))

    def trace_ao(self, spp=16, radius=0.2 * 1.457):
        """ao_layer.compute_ao (renderer.rs:688) replaced by ray-traced AO with XeGTAO's I/O contract"""
        check(self._L.art_trace_ao(self._ctx, spp, radius))

    def read_ao(self):
        w, h = self.extent
        a = np.empty((h, w), np.uint32)
        check(self._L.art_read_ao(self._ctx, _ptr(a), a.nbytes))
        return a

    def present(self):
        """tonemap_layer.present (renderer.rs:566-615): pack like the reference's images, then LPM tonemap to BGRA8"""
        check(self._L.art_present(self._ctx))

    def read_present(self):
        w, h = self.extent
        a = np.empty((h, w, 4), np.uint8)
        check(self._L.art_read_present(self._ctx, _ptr(a), a.nbytes))
        return a

    def read_packed(self):
        w, h = self.extent
        c, n, d = np.empty((h, w), np.uint32), np.empty((h, w), np.uint32), np.empty((h, w), np.uint16)
        check(self._L.art_read_packed(self._ctx, _ptr(c), _ptr(n), _ptr(d)))
        return c, n, d

    def render_frame(self, sync=True):  # renderer.rs:371
        self.update_models_status()
        self.upload_state()
        self.trace()
        if sync:
            self.sync()

    def sync(self):
        check(self._L.art_sync(self._ctx))

    # outputs (vk_rt_lightning_shadows.rs:161-183)
    def read_color(self):
        w, h = self.extent
        a = np.empty((h, w, 4), np.float32)
        check(self._L.art_read_color(self._ctx, _ptr(a), a.nbytes))
        return a

    def read_depth(self):
        w, h = self.extent
        a = np.empty((h, w), np.float32)
        check(self._L.art_read_depth(self._ctx, _ptr(a), a.nbytes))
        return a

    def read_normal(self):
        w, h = self.extent
        a = np.empty((h, w, 4), np.float32)
        check(self._L.art_read_normal(self._ctx, _ptr(a), a.nbytes))
        return a

    def device_color(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(self._L.art_device_color(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def _dev(self, which):
        """(device pointer, bytes) of the latest frame's depth / normal output (art_device_depth / art_device_normal)"""
        p, n = C.c_void_p(), C.c_size_t()
        check(getattr(self._L, "art_device_" + which)(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def device_color_tiles(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(self._L.art_device_color_tiles(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bind_color_tiles(self, slot, dev_ptr, nbytes):
        check(self._L.art_bind_color_tiles(self._ctx, slot, C.c_void_p(dev_ptr) if dev_ptr else None, nbytes))

    def bind_color_tiles_pair(self, slot, dev_even, dev_odd, nbytes):
        """two tile buffers per ring slot, written alternately: the next frame of a slot does not wait for the exchange of the previous one"""
        check(self._L.art_bind_color_tiles_pair(self._ctx, slot, C.c_void_p(dev_even), C.c_void_p(dev_odd), nbytes))

    def bind_color_tiles_ring(self, slot, dev_ptrs, nbytes):
        """len(dev_ptrs) tile buffers per ring slot, written in turn (one per trip round the frame ring)"""
        arr = (C.c_void_p * len(dev_ptrs))(*dev_ptrs)
        check(self._L.art_bind_color_tiles_ring(self._ctx, slot, arr, len(dev_ptrs), nbytes))

    def set_graph_mode(self, on):
        check(self._L.art_set_graph_mode(self._ctx, int(bool(on))))

    def set_frames_per_launch(self, n):
        """n frames per art_trace launch (fused frame); a ring slot then holds n frames"""
        check(self._L.art_set_frames_per_launch(self._ctx, n))
        self.frames_per_launch = n

    def set_camera_batch(self, cameras):
        """one Camera per frame of a launch"""
        arr = (ArtCamera * len(cameras))()
        for i, cam in enumerate(cameras):
            C.memmove(C.byref(arr[i]), C.byref(cam.update_host_buffer()), C.sizeof(ArtCamera))
        check(self._L.art_set_camera_batch(self._ctx, arr, len(cameras)))

    def set_read_frame(self, b):
        check(self._L.art_set_read_frame(self._ctx, b))

    def frames_done(self, first, count):
        """host-side, non-blocking: have frames [first, first + count) (art_trace order, from 0) all finished?"""
        d = C.c_int32()
        check(self._L.art_frames_done(self._ctx, first, count, C.byref(d), None))
        return bool(d.value)

    def frames_traced(self):
        d, n = C.c_int32(), C.c_uint64()
        check(self._L.art_frames_done(self._ctx, 0, 0, C.byref(d), C.byref(n)))
        return n.value

    def frames_in_flight(self):
        f, nxt = C.c_uint32(), C.c_uint32()
        check(self._L.art_frames_in_flight(self._ctx, C.byref(f), C.byref(nxt)))
        return f.value, nxt.value

    def stream_wait_frame(self, hip_stream_ptr):
        check(self._L.art_stream_wait_frame(self._ctx, C.c_void_p(hip_stream_ptr)))

    def trace_for_stream(self, hip_stream_ptr):
        """trace() + stream_wait_frame() in one call; returns the ring slot the frame took"""
        k = C.c_uint32()
        check(self._L.art_trace_for_stream(self._ctx, C.c_void_p(hip_stream_ptr), C.byref(k)))
        return k.value

    def wait_external_event(self, hip_event_ptr):
        check(self._L.art_wait_external_event(self._ctx, C.c_void_p(hip_event_ptr)))

    def layout(self) -> dict:
        lay = _lib.ArtLayout()
        check(self._L.art_get_layout(self._ctx, C.byref(lay)))
        return {n: getattr(lay, n) for n, _ in lay._fields_ if n != "reserved"}

    def timestamp_mark(self, which):
        """device timestamp behind the most recently traced frame (mark 0 or 1)"""
        check(self._L.art_timestamp_mark(self._ctx, which))

    def timestamp_elapsed_ms(self):
        ms = C.c_float()
        check(self._L.art_timestamp_elapsed(self._ctx, C.byref(ms)))
        return ms.value

    def collect_timings(self):
        sums = (C.c_float * 5)()
        n = C.c_uint32()
        check(self._L.art_collect_timings(self._ctx, sums, C.byref(n)))
        names = ("primary_ms", "shade_ms", "shadow_ms", "accumulate_ms", "frame_ms")
        k = max(1, n.value)
        return {nm: sums[i] / k for i, nm in enumerate(names)}, n.value

    def read_color_tiles(self):
        _, padded = self.shard_tile_count()
        a = np.empty((padded, 32, 32), np.uint32) if self.packed_tiles else np.empty((padded, 32, 32, 3), np.float32)   # RGB32F: the colour without its constant alpha
        check(self._L.art_read_color_tiles(self._ctx, _ptr(a), a.nbytes))
        return a

    def shard_tile_count(self):
        o, pd = C.c_uint32(), C.c_uint32()
        check(self._L.art_shard_tile_count(self._ctx, C.byref(o), C.byref(pd)))
        return o.value, pd.value

    def untile_gathered(self, gathered_dev_ptr, shard_count, frame_dev_ptr=None, hip_stream_ptr=None, shard_stride_tiles=None, n_frames=None):
        if n_frames is not None:   # several consecutive ring slots in one launch; frame_dev_ptr: n_frames images back to back
            check(self._L.art_untile_gathered_frames(self._ctx, C.c_void_p(gathered_dev_ptr), shard_count, shard_stride_tiles, n_frames,
                                                     C.c_void_p(frame_dev_ptr) if frame_dev_ptr else None, C.c_void_p(hip_stream_ptr) if hip_stream_ptr else None))
            return
        if shard_stride_tiles is not None:
            check(self._L.art_untile_gathered_strided(self._ctx, C.c_void_p(gathered_dev_ptr), shard_count, shard_stride_tiles,
                                                      C.c_void_p(frame_dev_ptr) if frame_dev_ptr else None, C.c_void_p(hip_stream_ptr) if hip_stream_ptr else None))
            return
        check(self._L.art_untile_gathered(self._ctx, C.c_void_p(gathered_dev_ptr), shard_count, C.c_void_p(frame_dev_ptr) if frame_dev_ptr else None,
                                          C.c_void_p(hip_stream_ptr) if hip_stream_ptr else None))

    def stats(self) -> dict:
        st = ArtStats()
        check(self._L.art_get_stats(self._ctx, C.byref(st)))
        return st.as_dict()

    # parity / debug surface
    def read_hits(self):
        w, h = self.extent
        tuv = np.empty((h, w, 4), np.float32)
        ids = np.empty((h, w, 2), np.int32)
        check(self._L.art_read_hits(self._ctx, _ptr(tuv), _ptr(ids), w * h))
        return tuv, ids

    def read_shadow_bits(self):
        w, h = self.extent
        b = np.empty((h, w), np.uint32)
        check(self._L.art_read_shadow_bits(self._ctx, _ptr(b), w * h))
        return b

    def query_closest(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        tuv = np.zeros((n, 4), np.float32)
        ids = np.zeros((n, 2), np.int32)
        check(self._L.art_query_closest(self._ctx, _ptr(rays), n, _ptr(tuv), _ptr(ids)))
        return tuv, ids

    def query_any(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        hit = np.zeros(n, np.uint8)
        check(self._L.art_query_any(self._ctx, _ptr(rays), n, _ptr(hit)))
        return hit

    def get_lbvh(self):
        T = self.stats()["num_triangles"]
        NI = max(T - 1, 0)
        out = dict(leaf_gid=np.zeros(T, np.uint32), keys=np.zeros(T, np.uint64), child=np.zeros((NI, 2), np.int32), node_lo=np.zeros((NI, 3), np.float32),
                   node_hi=np.zeros((NI, 3), np.float32), leaf_lo=np.zeros((T, 3), np.float32), leaf_hi=np.zeros((T, 3), np.float32))
        check(self._L.art_get_lbvh(self._ctx, *[_ptr(out[k]) for k in ("leaf_gid", "keys", "child", "node_lo", "node_hi", "leaf_lo", "leaf_hi")]))
        return out

    def get_traversal_tree(self):
        """the tree the walks use over get_lbvh()'s leaves (binned SAH by default)"""
        NI = max(self.stats()["num_triangles"] - 1, 0)
        out = dict(child=np.zeros((NI, 2), np.int32), node_lo=np.zeros((NI, 3), np.float32), node_hi=np.zeros((NI, 3), np.float32))
        check(self._L.art_get_traversal_tree(self._ctx, _ptr(out["child"]), _ptr(out["node_lo"]), _ptr(out["node_hi"])))
        return out

    def get_wide_nodes(self):
        """the 4-wide collapse the walks read: (n, 16) uint32 quantised records and (n, 32) uint32 float-box records (art_get_wide_nodes)"""
        n = C.c_uint32()
        check(self._L.art_get_wide_nodes(self._ctx, None, None, 0, C.byref(n)))
        q, f = np.zeros((n.value, 16), np.uint32), np.zeros((n.value, 32), np.uint32)
        check(self._L.art_get_wide_nodes(self._ctx, _ptr(q), _ptr(f), n.value, C.byref(n)))
        return q, f


def mgpu_shard(rank, world, dedicated=False):
    """(shard_rank, shard_count) the context of `rank` is created with (art_mgpu_shard)"""
    sr, sc = C.c_uint32(), C.c_uint32()
    check(_lib.load().art_mgpu_shard(rank, world, _lib.ART_MGPU_DEDICATED if dedicated else _lib.ART_MGPU_SHARED, C.byref(sr), C.byref(sc)))
    return sr.value, sc.value


def mgpu_unique_id() -> bytes:
    """rank 0: the job's 128-byte RCCL id, to be handed to the other ranks"""
    buf = (C.c_uint8 * _lib.ART_MGPU_ID_BYTES)()
    check(_lib.load().art_mgpu_unique_id(buf))
    return bytes(buf)


class MultiGpu:
    """The sharded frame behind the C ABI (art_mgpu_*): this rank's share traced, the tiles of a group of launches gathered by RCCL -- to rank 0
    (one gather per group), or with `spread` frame f to rank f mod world (one grouped send / receive per group) -- and un-tiled by one launch.
    `exchange` (a Python function (send_ptr, nbytes, recv_ptr, root, stream_ptr) -> None: every rank's nbytes, rank order, into recv_ptr on `root`)
    replaces RCCL for rehearsals on one GPU."""

    def __init__(self, renderer: "Renderer", rank, world, unique_id: bytes = None, dedicated=False, launches_per_gather=0, tile_buffers=0, exchange=None, spread=False):
        self._L = _lib.load()
        self.renderer, self.rank, self.world = renderer, rank, world
        cfg = _lib.ArtMgpuConfig(rank=rank, world=world, compositor=_lib.ART_MGPU_DEDICATED if dedicated else _lib.ART_MGPU_SHARED,
                                 launches_per_gather=launches_per_gather, tile_buffers=tile_buffers, transport=_lib.ART_MGPU_HOST_EXCHANGE if exchange else _lib.ART_MGPU_RCCL,
                                 roots=_lib.ART_MGPU_ROOT_SPREAD if spread else _lib.ART_MGPU_ROOT_RANK0)
        self._cb = None
        if exchange:
            def _cb(user, send, nbytes, recv, root, stream):
                try:
                    exchange(send, nbytes, recv, root, stream)
                    return 0
                except Exception:   # nothing may unwind into the C caller
                    import traceback
                    traceback.print_exc()
                    return -1
            self._cb = _lib.ArtMgpuExchangeFn(_cb)   # kept alive as long as the object
            cfg.exchange = self._cb
        idbuf = (C.c_uint8 * _lib.ART_MGPU_ID_BYTES)(*unique_id) if unique_id is not None else None
        self._h = C.c_void_p()
        check(self._L.art_mgpu_create(renderer._ctx, C.byref(cfg), idbuf, C.byref(self._h)))

    def trace(self):
        check(self._L.art_mgpu_trace(self._h))

    def flush(self):
        check(self._L.art_mgpu_flush(self._h))

    def pending(self):
        """(groups of launches whose exchange has not been submitted yet, launches of the group still open): exchanges are submitted lazily, from later
        launches or the flush"""
        g, n = C.c_uint32(), C.c_uint32()
        check(self._L.art_mgpu_pending(self._h, C.byref(g), C.byref(n)))
        return g.value, n.value

    def assert_quiescent(self, what="a control-plane collective"):
        """The rule a host with a control plane of its own must keep (a hang of round 2: rank 0 sat in the data gather of a queued group while the others
        had entered a broadcast of the control plane on the same gloo group): nothing of the data path may be outstanding when the ranks meet anywhere
        else -- flush() first."""
        g, n = self.pending()
        if g or n:
            raise RuntimeError(f"{what} while {g} group(s) of launches are queued for their exchange and {n} launch(es) wait in an open group: call flush() first")

    def counts(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint32()
        check(self._L.art_mgpu_counts(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(launches_traced=a.value, gathers=b.value, launches_per_gather=c.value)

    def read_frame(self):
        """a root (rank 0; every rank with spread roots), after flush(): the most recent frame it assembled -- [h, w, 4] float32, or [h, w] uint32 B10G11R11 words with packed tiles"""
        w, h = self.renderer.extent
        a = np.empty((h, w), np.uint32) if self.renderer.packed_tiles else np.empty((h, w, 4), np.float32)
        check(self._L.art_mgpu_read_frame(self._h, _ptr(a), a.nbytes))
        return a

    def close(self):
        if getattr(self, "_h", None):
            self._L.art_mgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def renderer_for_scene(scene, extent, n_lights=None, **kw) -> Renderer:
    """Convenience used by tests and bench: the main.rs:23-66 sequence for a synthetic scene."""
    r = Renderer(extent, **kw)
    r.add_model(scene.primitives)
    cam = r.camera_mut()
    cam.set_pos(scene.camera["pos"])
    cam.set_dir(scene.camera["dir"])
    cam.set_fovy(scene.camera["fovy"])
    cam.set_znear(scene.camera["znear"])
    cam.set_zfar(scene.camera["zfar"])
    for d in (scene.lights if n_lights is None else scene.lights[:n_lights]):
        r.lights_mut().push_dict(d)
    r.prepare_first_frame()
    return r
