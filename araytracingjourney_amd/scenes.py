"""Deterministic synthetic scenes in the reference's mesh layout.

The real assets of the reference (`assets/models/Sponza.glb`, ...) are missing blobs and there is no network,
so the BASELINE.json configs run on procedural stand-ins.  Every primitive is emitted exactly as the reference's
GLB reader would hand it to the renderer:

* vertices: 48-byte interleave pos3 / uv2 / normal3 / tangent4 (`model_reader.rs:22-35`,
  `gltf_model_reader.rs:176-199`),
* indices u16 when the primitive has <= 65535 vertices, else u32 (`vk_model.rs:142-150`),
* a 3-layer RGBA8 texture array: 0 albedo, 1 ORM, 2 normal (`model_reader.rs:14-19`),
* positions normalised into the unit ball (`gltf_model_reader.rs:415-460`), then the model matrix
  (`main.rs:30-36`: uniform scale 2 for Sponza).

No random-number generator is used: all variation comes from an integer lattice hash, so the output is the same
on every machine and numpy version.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------------- data model
@dataclass
class Primitive:
    verts: np.ndarray  # [nv, 12] float32
    indices: np.ndarray  # [n_idx] uint16 | uint32
    tex: np.ndarray  # [3, th, tw, 4] uint8
    model: np.ndarray = field(default_factory=lambda: np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=F32))

    @property
    def n_tris(self) -> int:
        return self.indices.size // 3


@dataclass
class Scene:
    name: str
    primitives: list
    camera: dict  # pos, dir, fovy, znear, zfar
    lights: list  # dicts: kind + parameters (see Lights in renderer.py)

    @property
    def n_tris(self) -> int:
        return sum(p.n_tris for p in self.primitives)


# --------------------------------------------------------------------------------------------- hash noise
def _hash_u32(ix, iy, seed):
    x = (np.asarray(ix).astype(np.uint64) * np.uint64(374761393) + np.asarray(iy).astype(np.uint64) * np.uint64(668265263)
         + np.uint64(seed) * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    x = ((x ^ (x >> np.uint64(13))) * np.uint64(1274126177)) & np.uint64(0xFFFFFFFF)
    x = ((x ^ (x >> np.uint64(16))) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    x = x ^ (x >> np.uint64(15))
    return x


def _lattice(ix, iy, seed):
    return (_hash_u32(ix, iy, seed) & np.uint64(0xFFFFFF)).astype(np.float64) / float(0x1000000)


def value_noise(x, y, seed, period=None):
    """Smooth value noise in [0,1); `period` (int) makes it tile."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    x0 = np.floor(x)
    y0 = np.floor(y)
    fx = x - x0
    fy = y - y0
    sx = fx * fx * (3 - 2 * fx)
    sy = fy * fy * (3 - 2 * fy)
    ix0 = x0.astype(np.int64)
    iy0 = y0.astype(np.int64)
    ix1 = ix0 + 1
    iy1 = iy0 + 1
    if period:
        ix0, ix1, iy0, iy1 = ix0 % period, ix1 % period, iy0 % period, iy1 % period
    else:
        ix0, ix1, iy0, iy1 = ix0 & 0xFFFF, ix1 & 0xFFFF, iy0 & 0xFFFF, iy1 & 0xFFFF
    a = _lattice(ix0, iy0, seed)
    b = _lattice(ix1, iy0, seed)
    c = _lattice(ix0, iy1, seed)
    d = _lattice(ix1, iy1, seed)
    return (a * (1 - sx) + b * sx) * (1 - sy) + (c * (1 - sx) + d * sx) * sy


def fbm(x, y, seed, octaves=3, period=None):
    out = 0.0
    amp = 0.5
    f = 1.0
    for o in range(octaves):
        out = out + amp * value_noise(np.asarray(x) * f, np.asarray(y) * f, seed + 101 * o, None if period is None else int(period * f))
        amp *= 0.5
        f *= 2.0
    return out


# --------------------------------------------------------------------------------------------- textures
def make_texture(seed, size, base_rgb, metallic, rough_lo=0.2, rough_hi=0.9, bump=0.35):
    """3-layer RGBA8 array [3, size, size, 4]: albedo, ORM (G roughness, B metallic), tangent-space normal."""
    yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    per = 8
    u = xx * (per / size)
    v = yy * (per / size)
    n1 = fbm(u, v, seed, 3, per)
    n2 = fbm(u, v, seed + 7, 3, per)
    tex = np.zeros((3, size, size, 4), dtype=np.uint8)
    base = np.asarray(base_rgb, dtype=np.float64)
    alb = np.clip(base[None, None, :] * (0.55 + 0.6 * n1[..., None]), 0.02, 1.0)
    tex[0, ..., :3] = np.round(alb * 255).astype(np.uint8)
    tex[0, ..., 3] = 255
    rough = rough_lo + (rough_hi - rough_lo) * np.clip(n2 * 1.4, 0, 1)
    tex[1, ..., 0] = 255
    tex[1, ..., 1] = np.round(rough * 255).astype(np.uint8)
    tex[1, ..., 2] = 255 if metallic else 0
    tex[1, ..., 3] = 255
    h = n1
    dx = (np.roll(h, -1, axis=1) - np.roll(h, 1, axis=1)) * bump * size / per
    dy = (np.roll(h, -1, axis=0) - np.roll(h, 1, axis=0)) * bump * size / per
    nz = np.ones_like(h)
    ln = np.sqrt(dx * dx + dy * dy + nz * nz)
    nrm = np.stack([-dx / ln, -dy / ln, nz / ln], axis=-1)
    tex[2, ..., :3] = np.round((nrm * 0.5 + 0.5) * 255).astype(np.uint8)
    tex[2, ..., 3] = 255
    return tex


def constant_texture(rgb, rough_g=128, metal_b=0, size=4):
    tex = np.zeros((3, size, size, 4), dtype=np.uint8)
    tex[0, ..., :3] = np.asarray(rgb, dtype=np.uint8)
    tex[0, ..., 3] = 255
    tex[1, ...] = np.array([255, rough_g, metal_b, 255], dtype=np.uint8)
    tex[2, ...] = np.array([128, 128, 255, 255], dtype=np.uint8)
    return tex


# --------------------------------------------------------------------------------------------- mesh helpers
class MeshBuilder:
    def __init__(self):
        self.v = []
        self.i = []
        self.nv = 0

    def add(self, pos, uv, nrm, tan, idx):
        pos = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
        n = pos.shape[0]
        vert = np.zeros((n, 12), dtype=np.float64)
        vert[:, 0:3] = pos
        vert[:, 3:5] = np.asarray(uv, dtype=np.float64).reshape(-1, 2)
        vert[:, 5:8] = np.asarray(nrm, dtype=np.float64).reshape(-1, 3)
        vert[:, 8:12] = np.asarray(tan, dtype=np.float64).reshape(-1, 4)
        self.v.append(vert)
        self.i.append(np.asarray(idx, dtype=np.int64).reshape(-1) + self.nv)
        self.nv += n

    def finish(self, tex, model=None) -> Primitive:
        v = np.concatenate(self.v, axis=0).astype(F32)
        i = np.concatenate(self.i, axis=0)
        idx = i.astype(np.uint16) if v.shape[0] <= 65535 else i.astype(np.uint32)
        p = Primitive(np.ascontiguousarray(v), np.ascontiguousarray(idx), tex)
        if model is not None:
            p.model = np.asarray(model, dtype=F32)
        return p


def _normalize(a):
    a = np.asarray(a, dtype=np.float64)
    return a / np.maximum(np.linalg.norm(a, axis=-1, keepdims=True), 1e-30)


def grid_indices(nu, nv):
    """Triangle indices of an (nu+1) x (nv+1) vertex grid (row-major in v then u)."""
    a = (np.arange(nv)[:, None] * (nu + 1) + np.arange(nu)[None, :]).reshape(-1)
    b = a + 1
    c = a + (nu + 1)
    d = c + 1
    return np.stack([a, b, d, a, d, c], axis=1).reshape(-1)


def parametric_patch(mb: MeshBuilder, fn, nu, nv, uv_scale=(1.0, 1.0), handed=1.0, flip=False):
    """fn(u, v) -> positions for u,v in [0,1]; normals/tangents from finite differences."""
    us = np.linspace(0.0, 1.0, nu + 1)
    vs = np.linspace(0.0, 1.0, nv + 1)
    U, V = np.meshgrid(us, vs, indexing="xy")
    U = U.reshape(-1)
    V = V.reshape(-1)
    e = 1e-4
    P = fn(U, V)
    Pu = fn(np.clip(U + e, 0, 1), V) - fn(np.clip(U - e, 0, 1), V)
    Pv = fn(U, np.clip(V + e, 0, 1)) - fn(U, np.clip(V - e, 0, 1))
    T = _normalize(Pu)
    N = _normalize(np.cross(Pu, Pv))
    if flip:
        N = -N
    T = _normalize(T - N * np.sum(T * N, axis=1, keepdims=True))
    tan = np.concatenate([T, np.full((T.shape[0], 1), handed)], axis=1)
    uv = np.stack([U * uv_scale[0], V * uv_scale[1]], axis=1)
    mb.add(P, uv, N, tan, grid_indices(nu, nv))


def quad(mb: MeshBuilder, p0, du, dv, nu=1, nv=1, uv_scale=(1.0, 1.0), handed=1.0, flip=False):
    p0 = np.asarray(p0, dtype=np.float64)
    du = np.asarray(du, dtype=np.float64)
    dv = np.asarray(dv, dtype=np.float64)
    parametric_patch(mb, lambda u, v: p0[None, :] + u[:, None] * du[None, :] + v[:, None] * dv[None, :], nu, nv, uv_scale, handed, flip)


def box(mb: MeshBuilder, lo, hi, handed=1.0, inward=False):
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    d = hi - lo
    ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    # each face: origin, du, dv with du x dv = outward normal
    faces = [
        (lo + ex, ez, ey),  # +x
        (lo, ey, ez),  # -x
        (lo + ey, ez, ex),  # +y
        (lo, ex, ez),  # -y
        (lo + ez, ex, ey),  # +z
        (lo, ey, ex),  # -z
    ]
    for o, a, b in faces:
        quad(mb, o, a, b, 1, 1, (1.0, 1.0), handed, flip=inward)


def icosphere(level):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = _normalize(np.array(v, dtype=np.float64))
    f = np.array(f, dtype=np.int64)
    for _ in range(level):
        edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        es = np.sort(edges, axis=1)
        key = es[:, 0] * (v.shape[0] + 1) + es[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        mids = _normalize((v[uniq // (v.shape[0] + 1)] + v[uniq % (v.shape[0] + 1)]) * 0.5)
        base = v.shape[0]
        v = np.concatenate([v, mids], axis=0)
        n = f.shape[0]
        ab, bc, ca = base + inv[:n], base + inv[n:2 * n], base + inv[2 * n:]
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        f = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)], axis=0)
    return v, f


def displaced_sphere(mb: MeshBuilder, centre, radius, level, seed, amp, handed=1.0):
    v, f = icosphere(level)
    lon = np.arctan2(v[:, 2], v[:, 0])
    lat = np.arcsin(np.clip(v[:, 1], -1, 1))
    # displacement from 3 projected noise lookups (seamless enough, deterministic)
    disp = (fbm(v[:, 0] * 3 + 7, v[:, 1] * 3 + 3, seed, 3) + fbm(v[:, 1] * 3 + 11, v[:, 2] * 3 + 5, seed + 1, 3) + fbm(v[:, 2] * 3 + 2, v[:, 0] * 3 + 9, seed + 2, 3)) / 3.0
    r = radius * (1.0 + amp * (disp - 0.45))
    P = np.asarray(centre, dtype=np.float64)[None, :] + v * r[:, None]
    # smooth normals from the faces
    fn = np.cross(P[f[:, 1]] - P[f[:, 0]], P[f[:, 2]] - P[f[:, 0]])
    N = np.zeros_like(P)
    for k in range(3):
        np.add.at(N, f[:, k], fn)
    N = _normalize(N)
    N = np.where(np.sum(N * v, axis=1, keepdims=True) < 0, -N, N)
    ref = np.where(np.abs(N[:, 1:2]) < 0.95, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    T = _normalize(np.cross(ref, N))
    tan = np.concatenate([T, np.full((T.shape[0], 1), handed)], axis=1)
    uv = np.stack([(lon / (2 * math.pi) + 0.5) * 4.0, (lat / math.pi + 0.5) * 4.0], axis=1)
    mb.add(P, uv, N, tan, f.reshape(-1))


def ngon_column(mb: MeshBuilder, base, radius, height, sides, rings, handed=1.0):
    base = np.asarray(base, dtype=np.float64)

    def shaft(u, v):
        ang = u * 2 * math.pi
        prof = radius * (1.0 - 0.12 * v + 0.05 * np.cos(ang * 8) * 0.2)
        return np.stack([base[0] + prof * np.cos(ang), base[1] + v * height, base[2] + prof * np.sin(ang)], axis=1)

    parametric_patch(mb, shaft, sides, rings, (2.0, 4.0), handed, flip=True)
    # capital and plinth
    c = 1.55 * radius
    box(mb, base + np.array([-c, height, -c]), base + np.array([c, height + 0.35 * radius * 2, c]), handed)
    box(mb, base + np.array([-c, -0.02, -c]), base + np.array([c, 0.3 * radius * 2, c]), handed)


def arch(mb: MeshBuilder, p0, p1, y, rise, thick, depth, segs, handed=1.0):
    """Half-ring arch between column tops p0 and p1 (xz positions), springing at height y."""
    p0 = np.asarray(p0, dtype=np.float64)
    p1 = np.asarray(p1, dtype=np.float64)
    mid = (p0 + p1) * 0.5
    half = np.linalg.norm(p1 - p0) * 0.5
    ax = (p1 - p0) / (2 * half)
    side = np.array([-ax[1], ax[0]])

    def ring(rad_scale, z_off, flip):
        def fn(u, v):
            ang = math.pi * (1.0 - u)
            r = half * rad_scale
            px = mid[0] + ax[0] * r * np.cos(ang) + side[0] * (v - 0.5) * depth * z_off
            pz = mid[1] + ax[1] * r * np.cos(ang) + side[1] * (v - 0.5) * depth * z_off
            py = y + rise * rad_scale * np.sin(ang)
            return np.stack([px, py, pz], axis=1)
        parametric_patch(mb, fn, segs, 2, (3.0, 1.0), handed, flip)

    ring(1.0 - thick, 1.0, False)
    ring(1.0, 1.0, True)


# --------------------------------------------------------------------------------------------- normalisation
def normalize_into_unit_ball(prims):
    """gltf_model_reader.rs:415-460: divide every position by max |p| when that exceeds 1."""
    m = max(float(np.max(np.linalg.norm(p.verts[:, 0:3].astype(np.float64), axis=1))) for p in prims)
    if m > 1.0:
        inv = F32(1.0) / F32(m)
        for p in prims:
            p.verts[:, 0:3] = p.verts[:, 0:3] * inv
    return m


def scale_matrix(s):
    return np.array([s, 0, 0, 0, 0, s, 0, 0, 0, 0, s, 0], dtype=F32)


# --------------------------------------------------------------------------------------------- C1 cornell
def cornell() -> Scene:
    """BASELINE config 1: 34 triangles, 3 primitives, 1 point light."""
    white, red, green = MeshBuilder(), MeshBuilder(), MeshBuilder()
    # room [-1,1]^3, open towards -z (camera side); normals point into the room
    quad(white, (-1, -1, -1), (2, 0, 0), (0, 0, 2), flip=False)   # floor  y=-1, normal +y ... checked below
    quad(white, (-1, 1, -1), (0, 0, 2), (2, 0, 0))                # ceiling y=+1, normal -y
    quad(white, (-1, -1, 1), (2, 0, 0), (0, 2, 0), flip=True)     # back wall z=+1, normal -z
    quad(red, (-1, -1, -1), (0, 0, 2), (0, 2, 0), flip=True)      # left wall x=-1, normal +x
    quad(green, (1, -1, -1), (0, 2, 0), (0, 0, 2), flip=True)     # right wall x=+1, normal -x
    box(white, (-0.65, -1.0, -0.1), (-0.05, 0.2, 0.5))            # tall box
    box(white, (0.15, -1.0, -0.55), (0.7, -0.4, 0.0))             # short box
    prims = [white.finish(constant_texture((200, 200, 200))), red.finish(constant_texture((200, 40, 40))),
             green.finish(constant_texture((40, 200, 40)))]
    _fix_room_normals(prims[0], 12)  # first 12 vertices = the three room quads
    _fix_room_normals(prims[1], 4)
    _fix_room_normals(prims[2], 4)
    normalize_into_unit_ball(prims)
    cam = dict(pos=(0.0, 0.0, -0.95), dir=(0.0, 0.0, 1.0), fovy=math.pi / 2, znear=0.1, zfar=1000.0)
    lights = [dict(kind="point", pos=(0.0, 0.5, 0.0), color=(8.0, 8.0, 8.0), falloff=3.0, casts_shadows=True)]
    return Scene("cornell", prims, cam, lights)


def _fix_room_normals(p: Primitive, n):
    """Make the first n vertex normals point towards the room centre (origin) and re-orthogonalise tangents."""
    v = p.verts
    d = np.sum(v[:n, 5:8] * v[:n, 0:3], axis=1)
    flip = d > 0
    v[:n, 5:8][flip] *= -1


# --------------------------------------------------------------------------------------------- C2/C3/C5 sponza-like
SPONZA_SEED = 0x5A0A


def sponza_like(detail: float = 1.0) -> Scene:
    """Two-storey atrium stand-in for Sponza: ~262 144 triangles in 25 primitives at detail=1."""
    s = SPONZA_SEED
    d = detail

    def n(x, lo=1):
        return max(lo, int(round(x * d)))

    L, Wd, Hh = 15.0, 6.0, 6.0  # half length (x), half width (z), half height (y)
    aisle = 2.5                 # side-aisle depth; nave is |z| < Wd - aisle
    prims = []
    handed_cycle = [1.0] * 9 + [-1.0]  # 10 % left-handed tangent frames

    def hd(k):
        return handed_cycle[k % 10]

    # 0 floor
    mb = MeshBuilder()
    parametric_patch(mb, lambda u, v: np.stack([-L + 2 * L * u, -Hh + 0.03 * fbm(u * 40, v * 16, s + 1), -Wd + 2 * Wd * v], 1), n(96), n(40), (4.0, 4.0), hd(0), flip=True)
    prims.append(mb.finish(make_texture(s + 10, 256, (0.62, 0.58, 0.5), False)))
    # 1 gallery floors (second storey, over the side aisles) incl. undersides
    mb = MeshBuilder()
    for sgn in (-1.0, 1.0):
        z0 = sgn * Wd
        z1 = sgn * (Wd - aisle)
        parametric_patch(mb, lambda u, v, z0=z0, z1=z1: np.stack([-L + 2 * L * u, np.full_like(u, 0.0), z0 + (z1 - z0) * v], 1), n(96), n(8), (4.0, 1.0), hd(1), flip=(sgn > 0))
        parametric_patch(mb, lambda u, v, z0=z0, z1=z1: np.stack([-L + 2 * L * u, np.full_like(u, -0.3), z0 + (z1 - z0) * v], 1), n(96), n(8), (4.0, 1.0), hd(1), flip=(sgn < 0))
    prims.append(mb.finish(make_texture(s + 11, 256, (0.55, 0.5, 0.45), False)))
    # 2 roof over the side aisles only (the nave is open to the sky, as in Sponza)
    mb = MeshBuilder()
    for sgn in (-1.0, 1.0):
        z0 = sgn * Wd
        z1 = sgn * (Wd - aisle)
        parametric_patch(mb, lambda u, v, z0=z0, z1=z1: np.stack([-L + 2 * L * u, np.full_like(u, Hh), z0 + (z1 - z0) * v], 1), n(96), n(8), (4.0, 1.0), hd(2), flip=(sgn < 0))
    prims.append(mb.finish(make_texture(s + 12, 256, (0.5, 0.45, 0.4), False)))
    # 3,4 long walls (bumpy masonry), 5 end walls
    for k, sgn in enumerate((-1.0, 1.0)):
        mb = MeshBuilder()
        parametric_patch(mb, lambda u, v, sgn=sgn: np.stack([-L + 2 * L * u, -Hh + 2 * Hh * v, sgn * (Wd + 0.05 * fbm(u * 60, v * 24, s + 20 + k))], 1), n(96), n(32), (4.0, 4.0), hd(3 + k), flip=(sgn > 0))
        prims.append(mb.finish(make_texture(s + 13 + k, 256, (0.66, 0.6, 0.52), False)))
    mb = MeshBuilder()
    for sgn in (-1.0, 1.0):
        parametric_patch(mb, lambda u, v, sgn=sgn: np.stack([np.full_like(u, sgn * L), -Hh + 2 * Hh * v, -Wd + 2 * Wd * u], 1), n(40), n(32), (4.0, 4.0), hd(5), flip=(sgn < 0))
    prims.append(mb.finish(make_texture(s + 15, 256, (0.6, 0.55, 0.5), False)))
    # 6..9 columns: 2 storeys x 2 rows x 14
    xs = np.linspace(-L + 1.2, L - 1.2, 14)
    col_r = 0.32
    for storey, y0 in enumerate((-Hh, 0.0)):
        for row, sgn in enumerate((-1.0, 1.0)):
            mb = MeshBuilder()
            for x in xs:
                ngon_column(mb, (x, y0, sgn * (Wd - aisle)), col_r, 4.6, n(16, 6), n(8, 2), hd(6 + storey * 2 + row))
            metallic = storey == 1 and row == 1
            prims.append(mb.finish(make_texture(s + 16 + storey * 2 + row, 256, (0.7, 0.66, 0.6) if not metallic else (0.9, 0.7, 0.3), metallic)))
    # 10,11 arches per storey
    for storey, y0 in enumerate((-Hh + 4.6 + 0.22, 0.0 + 4.6 + 0.22)):
        mb = MeshBuilder()
        for sgn in (-1.0, 1.0):
            for a, b in zip(xs[:-1], xs[1:]):
                arch(mb, (a, sgn * (Wd - aisle)), (b, sgn * (Wd - aisle)), y0, 0.9, 0.18, 0.5, n(16, 4), hd(10 + storey))
        prims.append(mb.finish(make_texture(s + 30 + storey, 256, (0.68, 0.62, 0.55), False)))
    # 12..23 twelve cloth banners hanging in the nave / from the galleries
    for c in range(12):
        mb = MeshBuilder()
        cx = -L + 2.5 + (2 * L - 5.0) * (c / 11.0)
        cz = (-1.0 if c % 2 else 1.0) * (Wd - aisle - 0.45)
        wdt, hgt = 1.7, 3.4
        ph = 0.9 * c

        def cloth(u, v, cx=cx, cz=cz, ph=ph, c=c):
            sag = 0.22 * np.sin(u * math.pi * 3 + ph) * (0.3 + v) + 0.12 * (fbm(u * 6, v * 6, s + 40 + c) - 0.5)
            return np.stack([cx + (u - 0.5) * wdt, 4.2 - v * hgt, cz + sag * (-1.0 if cz > 0 else 1.0)], 1)

        parametric_patch(mb, cloth, n(42), n(42), (2.0, 4.0), hd(12 + c))
        col = [(0.75, 0.12, 0.1), (0.1, 0.25, 0.7), (0.15, 0.55, 0.2)][c % 3]
        prims.append(mb.finish(make_texture(s + 50 + c, 256, col, False, 0.45, 0.9)))
    # 24 eight noise-displaced icospheres (dense clusters; > 65535 vertices => u32 indices)
    mb = MeshBuilder()
    lvl = 5 if d >= 0.75 else (4 if d >= 0.35 else 2)
    for k in range(8):
        cx = -L + 3.0 + (2 * L - 6.0) * (k / 7.0)
        cz = 1.1 * math.sin(k * 2.4)
        rad = 0.55 + 0.25 * ((k * 37) % 5) / 4.0
        displaced_sphere(mb, (cx, -Hh + rad * 1.05, cz), rad, lvl, s + 70 + k, 0.35, hd(k))
    prims.append(mb.finish(make_texture(s + 90, 256, (0.8, 0.8, 0.85), True, 0.2, 0.6)))
    assert len(prims) == 25
    normalize_into_unit_ball(prims)
    m = scale_matrix(2.0)  # main.rs:30-36
    for p in prims:
        p.model = m.copy()
    cam = dict(pos=(-1.2, 0.35, 0.0), dir=_unit((1.0, -0.05, 0.1)), fovy=math.pi / 2, znear=0.1, zfar=1000.0)
    lights = sponza_lights(1)
    return Scene("sponza_like", prims, cam, lights)


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    v = v / np.linalg.norm(v)
    return tuple(float(x) for x in v)


def sponza_lights(n_lights: int):
    """C2: one directional light.  C3: + point + the main.rs:42-49 spot + the main.rs:55-64 area light."""
    lights = [dict(kind="directional", dir=(-0.3, -1.0, -0.2), color=(3.0, 3.0, 3.0), casts_shadows=True)]
    if n_lights >= 4:
        lights.insert(0, dict(kind="point", pos=(0.0, 1.0, 0.0), color=(8.0, 8.0, 8.0), falloff=3.0, casts_shadows=True))
        lights.insert(1, dict(kind="spot", pos=(0.0, 1.5, 0.0), dir=(0.0, -1.0, 0.0), color=(13.6, 1.6, 22.2), falloff=3.0,
                              penumbra=math.radians(30.0), umbra=math.radians(45.0), casts_shadows=True))
        lights.append(dict(kind="area", pos=(-0.70, 0.77, 0.08), pos2=(-0.70, 0.77, -0.16), pos3=(-0.70, 0.90, -0.16), invert_normal=False,
                           color=(1.96 * 3, 0.06 * 3, 0.41 * 3), falloff=3.0, penumbra=math.radians(90.0), umbra=math.radians(90.0),
                           casts_shadows=True))
    return lights


# --------------------------------------------------------------------------------------------- C4 bistro-like
BISTRO_SEED = 0xB157


def bistro_like(detail: float = 1.0) -> Scene:
    """Street of 40 facades + 3000 small props (~2.8 M triangles, 120 primitives at detail=1)."""
    s = BISTRO_SEED
    d = detail

    def n(x, lo=1):
        return max(lo, int(round(x * math.sqrt(d))))

    prims = []
    street_len, half_w = 80.0, 5.0
    # 0 street, 1 pavements
    mb = MeshBuilder()
    parametric_patch(mb, lambda u, v: np.stack([-street_len / 2 + street_len * u, 0.02 * fbm(u * 200, v * 20, s + 1), -half_w + 2 * half_w * v], 1), n(400), n(50), (4.0, 4.0), 1.0, flip=True)
    prims.append(mb.finish(make_texture(s + 1, 128, (0.3, 0.3, 0.32), False)))
    # 40 facades (20 per side), each its own primitive: wall with window recesses via displacement
    for f in range(40):
        side = -1.0 if f % 2 else 1.0
        x0 = -street_len / 2 + (f // 2) * (street_len / 20.0)
        wdt = street_len / 20.0
        hgt = 7.0 + 5.0 * float(_lattice(f, 3, s))
        mb = MeshBuilder()

        def wall(u, v, x0=x0, wdt=wdt, hgt=hgt, side=side, f=f):
            win = (np.abs(np.sin(u * math.pi * 4)) > 0.55) & (np.abs(np.sin(v * math.pi * 5)) > 0.5)
            rec = np.where(win, 0.25, 0.0) + 0.06 * fbm(u * 30, v * 30, s + 100 + f)
            return np.stack([x0 + u * wdt, v * hgt, side * (half_w + rec)], 1)

        parametric_patch(mb, wall, n(150), n(150), (4.0, 4.0), 1.0 if f % 10 else -1.0, flip=(side > 0))
        # roof slab
        quad(mb, (x0, hgt, side * half_w), (wdt, 0, 0), (0, 0, side * 4.0), n(8), n(8), (2.0, 2.0), 1.0, flip=(side > 0))
        prims.append(mb.finish(make_texture(s + 200 + f, 128, (0.5 + 0.4 * float(_lattice(f, 1, s)), 0.45 + 0.3 * float(_lattice(f, 2, s)), 0.4), False)))
    # 3000 props in 78 primitives (icospheres / boxes scattered on the pavements)
    n_props = max(78, int(3000 * d))
    per = [n_props // 78 + (1 if i < n_props % 78 else 0) for i in range(78)]
    k = 0
    for g in range(78):
        mb = MeshBuilder()
        for _ in range(per[g]):
            px = -street_len / 2 + street_len * float(_lattice(k, 11, s))
            pz = (half_w - 0.3 - 1.4 * float(_lattice(k, 12, s))) * (1.0 if _hash_u32(k, 13, s) & np.uint64(1) else -1.0)
            r = 0.12 + 0.3 * float(_lattice(k, 14, s))
            if _hash_u32(k, 15, s) % np.uint64(3) == 0:
                box(mb, (px - r, 0.0, pz - r), (px + r, 2.2 * r, pz + r))
            else:   # 15 % level-3 icospheres (1280 triangles), the rest level 2 (320): ~2.8 M triangles in total
                displaced_sphere(mb, (px, r, pz), r, 3 if float(_lattice(k, 16, s)) < 0.15 else 2, s + 1000 + k, 0.3)
            k += 1
        prims.append(mb.finish(make_texture(s + 400 + g, 128, (0.3 + 0.6 * float(_lattice(g, 5, s)), 0.5, 0.3 + 0.5 * float(_lattice(g, 6, s))), g % 9 == 0)))
    # 1 more: awnings strip
    mb = MeshBuilder()
    for sgn in (-1.0, 1.0):
        parametric_patch(mb, lambda u, v, sgn=sgn: np.stack([-street_len / 2 + street_len * u, 3.2 - 0.5 * v + 0.08 * np.sin(u * 300), sgn * (half_w - 1.5 * v)], 1), n(600), n(12), (4.0, 1.0), 1.0, flip=(sgn < 0))
    prims.append(mb.finish(make_texture(s + 900, 128, (0.7, 0.15, 0.12), False)))
    assert len(prims) == 120, len(prims)
    normalize_into_unit_ball(prims)
    m = scale_matrix(2.0)
    for p in prims:
        p.model = m.copy()
    cam = dict(pos=(-1.85, 0.085, 0.0), dir=_unit((1.0, 0.02, 0.03)), fovy=math.pi / 2, znear=0.1, zfar=1000.0)
    lights = [dict(kind="directional", dir=(-0.25, -1.0, -0.35), color=(3.0, 3.0, 3.0), casts_shadows=True)]
    return Scene("bistro_like", prims, cam, lights)


def camera_path(scene: Scene, n: int):
    """n camera poses around the scene's own (bench.py --camera-path): a closed loop of small displacements with the view direction swinging
    along, so that consecutive frames walk different parts of the tree.  Pose 0 is the scene's camera."""
    p0, d0 = np.asarray(scene.camera["pos"], np.float64), np.asarray(scene.camera["dir"], np.float64)
    out = []
    for i in range(n):
        a = 2.0 * math.pi * i / n
        pos = p0 + np.array([0.22 * math.sin(a), 0.06 * (1.0 - math.cos(a)), 0.18 * (1.0 - math.cos(a))])
        d = d0 + np.array([0.0, 0.10 * math.sin(a), 0.35 * math.sin(a) - 0.15 * (1.0 - math.cos(a))])
        out.append(dict(scene.camera, pos=tuple(float(x) for x in pos), dir=_unit(d)))
    return out


def camera_walk(scene: Scene, n: int, fps: float = 6000.0):
    """n camera poses of the REFERENCE's own camera motion (bench.py --camera-walk): main.rs:80-105 moves the camera 0.002 units per millisecond of frame time along a view
    axis while a key is down (2 units / s) and turns it 0.002 rad per mouse count (:112-124; a brisk hand makes ~1 000 counts / s: 2 rad / s).  At `fps` frames per second -- the
    rate this library renders the scene at: what a render loop on top of it sees -- that is 2 / fps units and 2 / fps rad a frame.  A closed loop: n / 2 frames forward along the
    view direction while turning about the up axis, n / 2 frames back.  Pose 0 is the scene's camera."""
    p0, d0 = np.asarray(scene.camera["pos"], np.float64), np.asarray(_unit(scene.camera["dir"]), np.float64)
    step, turn = 2.0 / fps, 2.0 / fps
    out = []
    for i in range(n):
        k = min(i, n - i)
        yaw = turn * k
        d = np.array([d0[0] * math.cos(yaw) + d0[2] * math.sin(yaw), d0[1], -d0[0] * math.sin(yaw) + d0[2] * math.cos(yaw)])   # about the up axis (0, -1, 0) of vk_camera.rs:182-189
        out.append(dict(scene.camera, pos=tuple(float(x) for x in (p0 + d0 * (step * k))), dir=_unit(d)))
    return out


def from_glb(path, lights=None, camera=None) -> Scene:
    """A .glb through the C++ reader (art_glb_*: GltfModelReader of gltf_model_reader.rs) as a Scene, set up like the reference's main.rs:23-66:
    model matrix = uniform scale 2 (:30-36), the renderer's default camera (renderer.rs:222-231), one directional light unless given.
    Like VkModel (vk_model.rs:498-508) it needs positions, uvs, normals, tangents, indices and the albedo / ORM / normal textures."""
    from . import model_reader as mr
    r = mr.GltfModelReader(str(path), True, mr.COERCE_R8G8B8A8)
    data, infos = r.copy_model_data_to_ptr(mr.VERTICES | mr.TEX_COORDS | mr.NORMALS | mr.TANGENTS | mr.INDICES, mr.ALBEDO | mr.ORM | mr.NORMAL)
    prims = []
    for ci in infos:
        if ci.single_mesh_element_size != 48 or ci.image_layers != 3:
            raise ValueError("the GLB does not carry the 48-byte vertex + three texture layers the ray tracer reads")
        verts = data[ci.mesh_buffer_offset:ci.mesh_buffer_offset + ci.mesh_size].view(F32).reshape(-1, 12).copy()
        idx = data[ci.indices_buffer_offset:ci.indices_buffer_offset + ci.indices_size].view(np.uint16 if ci.single_index_size == 2 else np.uint32).copy()
        tex = data[ci.image_buffer_offset:ci.image_buffer_offset + ci.image_size].reshape(3, ci.image_height, ci.image_width, 4).copy()
        prims.append(Primitive(verts, idx, tex, scale_matrix(2.0)))
    r.close()
    cam = camera or dict(pos=(0.0, 0.0, 0.0), dir=(0.0, 0.0, 1.0), fovy=math.pi / 2, znear=0.1, zfar=1000.0)
    return Scene("glb:" + os.path.basename(str(path)), prims, cam, lights if lights is not None else sponza_lights(1))


def get_scene(name: str, detail: float = 1.0) -> Scene:
    if name == "cornell":
        return cornell()
    if name == "sponza_like":
        return sponza_like(detail)
    if name == "bistro_like":
        return bistro_like(detail)
    raise ValueError(f"unknown scene {name!r}")
