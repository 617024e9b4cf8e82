"""Independent numpy restatement of the reference's shading maths, written from the GLSL
(/root/reference/src/vk_renderer/shaders/rt_lightning_shadows/raytrace.rgen.glsl:106-199, light.glsl, brdfs.glsl),
NOT from the C oracle: it exists to catch transcription errors the oracle and the HIP kernels could share.
float64 throughout; one pixel at a time is fine at the sizes the tests use."""
import math

import numpy as np

PI = 3.14159265359


def normalize(v):
    return v / np.sqrt(np.dot(v, v))


def mix(x, y, a):
    return x * (1.0 - a) + y * a


def clamp(x, lo, hi):
    return min(max(x, lo), hi)


def texture(tex_layer, uv):
    """sampler2DArray layer, linear filter, REPEAT, LOD 0 (vk_rt_descriptor_set.rs:42-56)"""
    th, tw = tex_layer.shape[:2]
    x, y = uv[0] * tw - 0.5, uv[1] * th - 0.5
    x0, y0 = math.floor(x), math.floor(y)
    fx, fy = x - x0, y - y0
    t = lambda xx, yy: tex_layer[yy % th, xx % tw].astype(np.float64) / 255.0
    return (t(x0, y0) * (1 - fx) + t(x0 + 1, y0) * fx) * (1 - fy) + (t(x0, y0 + 1) * (1 - fx) + t(x0 + 1, y0 + 1) * fx) * fy


def D_GGX(roughness, NdotH):
    a = NdotH * roughness
    k = roughness / (1.0 - NdotH * NdotH + a * a)
    return k * k * (1.0 / PI)


def V_SmithGGXCorrelated_fast(roughness, NdotV, NdotL):
    return 0.5 / mix(2 * NdotL * NdotV, NdotL + NdotV, roughness)


def F_Schlick(F0, F90, x):
    return F0 + (F90 - F0) * (1.0 - x) ** 5.0


def Burley_diffuse_local_sss(roughness, NdotV, nc_NdotV, nc_NdotL, LdotH, ratio):
    F_SS90 = roughness * LdotH * LdotH
    F_SS = F_Schlick(1.0, F_SS90, nc_NdotL) * F_Schlick(1.0, F_SS90, nc_NdotV)
    f_ss = (1.0 / (nc_NdotV * nc_NdotL) - 0.5) * F_SS + 0.5
    local_sss = 1.25 * ratio * f_ss
    f90 = 0.5 + 2.0 * F_SS90
    diffuse = (1.0 - ratio) * F_Schlick(1.0, f90, nc_NdotL) * F_Schlick(1.0, f90, nc_NdotV)
    return NdotV * (diffuse + local_sss) * (1.0 / PI)


def compute_barycentric(a, b, c, p):
    v0, v1, v2 = b - a, c - a, p - a
    d00, d01, d11, d20, d21 = v0 @ v0, v0 @ v1, v1 @ v1, v2 @ v0, v2 @ v1
    denom = d00 * d11 - d01 * d01
    x = (d11 * d20 - d01 * d21) / denom
    y = (d00 * d21 - d01 * d20) / denom
    return np.array([x, y, 1 - x - y])


def closest_point_to_segment(p0, p1, p):
    v = p1 - p0
    t = clamp(((p - p0) @ v) / (v @ v), 0.0, 1.0)
    return p0 + t * v


def closest_point_to_triangle(p0, p1, p2, pt):
    b = compute_barycentric(p0, p1, p2, pt)
    if b[0] < 0:
        return closest_point_to_segment(p2, p0, pt)
    elif b[2] < 0:
        return closest_point_to_segment(p1, p2, pt)
    return pt


def get_unnormalized_L_vec(light, pos):
    t = light["type"]
    if t in (0, 1):
        return light["pos"] - pos
    if t == 2:
        return -light["dir"] * 10.0
    if t == 3:
        distance = light["dir"] @ light["area_pos2"] - light["dir"] @ pos
        cp = pos + distance * light["dir"]
        b = compute_barycentric(light["pos"], light["area_pos2"], light["area_pos3"], cp)
        if b[0] < 0:
            pos4 = light["pos"] - light["area_pos2"] + light["area_pos3"]
            c = closest_point_to_triangle(light["pos"], light["area_pos3"], pos4, cp)
        elif b[1] < 0:
            c = closest_point_to_segment(light["pos"], light["area_pos2"], cp)
        elif b[2] < 0:
            c = closest_point_to_segment(light["area_pos2"], light["area_pos3"], cp)
        else:
            c = cp
        return c - pos
    return np.ones(3)


def get_light_radiance(light, pos, L):
    rad = light["color"].copy()
    if light["type"] in (1, 3):
        theta = math.acos(clamp(light["dir"] @ (-L), -1.0, 1.0))
        with np.errstate(divide="ignore", invalid="ignore"):   # penumbra == umbra (main.rs:62) divides by zero, as in GLSL
            q = np.float64(theta - light["umbra"]) / np.float64(light["penumbra"] - light["umbra"])
        t = clamp(q, 0.0, 1.0)
        rad = rad * t ** 2.0
    if light["falloff"] > 0:
        dist = np.linalg.norm(light["pos"] - pos)
        rad = rad * max(1 - (dist / light["falloff"]) ** 2.0, 0.0) ** 2.0
    return rad


def light_from_record(rec):
    """rec: an 80-byte light record as a ctypes struct (either library's)"""
    f = lambda a: np.array(list(a), dtype=np.float64)
    return dict(pos=f(rec.pos), type=int(rec.type), dir=f(rec.dir), casts=bool(rec.casts_shadows), color=f(rec.color), falloff=float(rec.falloff_distance),
                area_pos2=f(rec.area_pos2), penumbra=float(rec.penumbra_angle), area_pos3=f(rec.area_pos3), umbra=float(rec.umbra_angle))


def shade_pixel(prim, tri, u, v, cam_view, cam_view_inv, camera_pos, lights, shadowed_bits):
    """prim: scenes.Primitive; cam_*: 4x4 numpy (row, col); returns (color3, depth, normal3, mask): mask bit 16 + i = a shadow ray towards light i is due,
    bit i = that light was found shadowed (only when shadowed_bits is a function that traces the ray)"""
    i0, i1, i2 = [int(x) for x in prim.indices[3 * tri:3 * tri + 3]]
    V = prim.verts.astype(np.float64)
    v0, v1, v2 = V[i0], V[i1], V[i2]
    bary = np.array([1.0 - u - v, u, v])
    M = np.vstack([prim.model.astype(np.float64).reshape(3, 4), [0, 0, 0, 1]])   # object -> world
    Minv = np.linalg.inv(M)
    interp = lambda a, b: v0[a:b] * bary[0] + v1[a:b] * bary[1] + v2[a:b] * bary[2]
    pos = interp(0, 3)
    world_pos = (M @ np.append(pos, 1.0))[:3]
    tex_coord = interp(3, 5)
    normal = normalize(interp(5, 8))
    world_normal = normalize((normal @ Minv[:3, :3]))            # normal * world_to_object
    tangent = normalize(interp(8, 11))
    world_tangent = normalize(M[:3, :3] @ tangent)
    world_tangent = normalize(world_tangent - (world_tangent @ world_normal) * world_normal)
    world_binormal = np.cross(world_normal, world_tangent) * v0[11]
    tbn = np.stack([world_tangent, world_binormal, world_normal], axis=1)
    N = normalize(texture(prim.tex[2], tex_coord)[:3] * 2.0 - 1.0)
    N = normalize(tbn @ N)
    albedo = texture(prim.tex[0], tex_coord)[:3] ** 2.2
    orm = texture(prim.tex[1], tex_coord)
    roughness, metallic = orm[1], orm[2]
    Vv = normalize(camera_pos - world_pos)
    F0 = mix(np.full(3, 0.04), albedo, metallic)
    cr = roughness * roughness
    nc_NdotV = N @ Vv
    NdotV = clamp(nc_NdotV, 1e-5, 1.0)
    rho = np.zeros(3)
    mask = 0
    for i, l in enumerate(lights):
        nn_L = get_unnormalized_L_vec(l, world_pos)
        L = normalize(nn_L)
        H = normalize(Vv + L)
        nc_NdotL = N @ L
        NdotL = clamp(nc_NdotL, 0.0, 1.0)
        NdotH = clamp(N @ H, 0.0, 1.0)
        LdotH = clamp(L @ H, 0.0, 1.0)
        Ks = F_Schlick(F0, 1.0, LdotH)
        Kd = (1.0 - metallic) * albedo
        rho_s = D_GGX(cr, NdotH) * V_SmithGGXCorrelated_fast(cr, NdotV, NdotL) * Ks
        rho_d = Kd * Burley_diffuse_local_sss(cr, NdotV, nc_NdotV, nc_NdotL, LdotH, 0.4)
        att = 1.0
        if l["casts"] and nc_NdotL > 0:
            mask |= 1 << (16 + i)
            # shadowed_bits: the bits someone else found, or a function (light index, origin, direction, tmax) -> bool that traces the shadow ray itself
            # (raytrace.rgen.glsl:165-181: origin world_pos, tmin 0.01, direction L, tmax length(nn_L))
            if callable(shadowed_bits):
                if shadowed_bits(i, world_pos, L, float(np.sqrt(nn_L @ nn_L))):
                    att = 0.05
                    mask |= 1 << i
            elif shadowed_bits >> i & 1:
                att = 0.05
        rho = rho + (rho_s + rho_d) * get_light_radiance(l, world_pos, L) * att * NdotL
    depth = -(cam_view @ np.append(world_pos, 1.0))[2]
    on = cam_view_inv[:3, :3].T @ N           # mat3(transpose(view_inv)) * N
    on[1:] = -on[1:]
    on = normalize(on) * 0.5 + 0.5
    return rho, depth, on, mask


# ---- geometry, from the same sources: camera block (vk_camera.rs:104-126, :182-193 with nalgebra's look_at_rh / Perspective3 as
# SURVEY.md 8a spells them out), ray generation (raytrace.rgen.glsl:78-88), and what traceRayEXT returns for opaque two-sided
# triangles -- the closest Moeller-Trumbore hit in (tmin, tmax) -- by brute force over every triangle, in float64 ------------------
def camera_matrices(pos, dir, aspect, fovy, znear, zfar):
    """-> view, view_inv, proj, proj_inv as 4x4 (row, col) float64"""
    eye = np.asarray(pos, np.float64)
    f = normalize(np.asarray(dir, np.float64))
    up = np.array([0.0, -1.0, 0.0])
    s = normalize(np.cross(f, up))
    u = np.cross(s, f)
    view = np.array([[s[0], s[1], s[2], -s @ eye], [u[0], u[1], u[2], -u @ eye], [-f[0], -f[1], -f[2], f @ eye], [0, 0, 0, 1.0]])
    c = 1.0 / math.tan(fovy / 2.0)
    proj = np.array([[c / aspect, 0, 0, 0], [0, c, 0, 0], [0, 0, (zfar + znear) / (znear - zfar), 2.0 * zfar * znear / (znear - zfar)], [0, 0, -1.0, 0]])
    return view, np.linalg.inv(view), proj, np.linalg.inv(proj)


def primary_ray(x, y, w, h, view_inv, proj_inv):
    pc = np.array([x + 0.5, y + 0.5])
    d = pc / np.array([w, h], np.float64) * 2.0 - 1.0
    origin = (view_inv @ np.array([0.0, 0.0, 0.0, 1.0]))[:3]
    target = proj_inv @ np.array([d[0], d[1], 1.0, 1.0])
    direction = (view_inv @ np.append(normalize(target[:3]), 0.0))[:3]
    return origin, direction


def world_triangles(primitives):
    """-> [T, 3, 3] world-space vertices, [T] primitive index, [T] triangle index inside its primitive"""
    tris, pid, tid = [], [], []
    for pi, p in enumerate(primitives):
        M = p.model.astype(np.float64).reshape(3, 4)
        P = p.verts[:, :3].astype(np.float64) @ M[:, :3].T + M[:, 3]
        idx = p.indices.astype(np.int64).reshape(-1, 3)
        tris.append(P[idx]); pid += [pi] * len(idx); tid += list(range(len(idx)))
    return np.concatenate(tris), np.array(pid), np.array(tid)


def closest_hit(o, d, tris, tmin=0.001, tmax=10000.0, eps=0.0):
    """two-sided Moeller-Trumbore against all triangles at once -> (index into tris or -1, t, u, v, margin): margin = how far inside its
    triangle the hit lies in barycentric units (a hit within ~1e-6 of an edge may legitimately go to the neighbour in another implementation)"""
    v0, e1, e2 = tris[:, 0], tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    p = np.cross(d, e2)
    det = np.einsum("ij,ij->i", e1, p)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / det
        tv = o - v0
        u = np.einsum("ij,ij->i", tv, p) * inv
        q = np.cross(tv, e1)
        v = (q @ d) * inv
        t = np.einsum("ij,ij->i", e2, q) * inv
    ok = (det != 0) & (u >= -eps) & (v >= -eps) & (u + v <= 1 + eps) & (t > tmin) & (t < tmax)   # eps: edges fattened (what makes shared edges watertight)
    if not ok.any():
        return -1, tmax, 0.0, 0.0, 0.0
    i = int(np.argmin(np.where(ok, t, np.inf)))
    return i, float(t[i]), float(u[i]), float(v[i]), float(min(u[i], v[i], 1.0 - u[i] - v[i]))


class BruteForce:
    """Every triangle of a scene in float64, for a brute-force witness on scenes of 10^5 .. 10^6 triangles (the bench scenes): no tree, no traversal order, no
    Morton keys, nothing the oracle's BVH could share a mistake with.  Rays that share an origin (a camera's) are tested in batches through the scalar triple
    products of Moeller-Trumbore written as matrix products -- det = d . (e2 x e1), u det = d . (e2 x tv), v det = d . (tv x e1), t det = e2 . (tv x e1) -- which is
    the same test in another order of operations (float64: the orders differ by ~1e-13)."""

    def __init__(self, primitives):
        self.tris, self.pid, self.tid = world_triangles(primitives)
        self.v0 = np.ascontiguousarray(self.tris[:, 0]); self.e1 = self.tris[:, 1] - self.v0; self.e2 = self.tris[:, 2] - self.v0
        self.n1 = np.cross(self.e2, self.e1)
        self.lo, self.hi = self.tris.min(1), self.tris.max(1)

    def closest_from(self, o, D, tmin, tmax, eps=0.0, chunk_elems=24_000_000):
        """rays o + t D[r]: -> (index [R] into the triangles or -1, t, u, v, margin) of the closest accepted hit of each"""
        tv = o - self.v0
        A, Q = np.cross(self.e2, tv), np.cross(tv, self.e1)
        tdet = np.einsum("ij,ij->i", self.e2, Q)
        R, T = len(D), len(self.v0)
        idx, tt, uu, vv, mm = np.full(R, -1), np.full(R, tmax), np.zeros(R), np.zeros(R), np.zeros(R)
        step = max(1, chunk_elems // T)
        for r0 in range(0, R, step):
            d = D[r0:r0 + step]
            det = d @ self.n1.T
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / det
                u = (d @ A.T) * inv
                v = (d @ Q.T) * inv
                t = tdet[None, :] * inv
            ok = (det != 0) & (u >= -eps) & (v >= -eps) & (u + v <= 1 + eps) & (t > tmin) & (t < tmax)
            t = np.where(ok, t, np.inf)
            i = np.argmin(t, axis=1)
            rr = np.arange(len(d))
            hit = np.isfinite(t[rr, i])
            sel = r0 + rr[hit]
            idx[sel] = i[hit]; tt[sel] = t[rr[hit], i[hit]]; uu[sel] = u[rr[hit], i[hit]]; vv[sel] = v[rr[hit], i[hit]]
            mm[sel] = np.minimum(np.minimum(uu[sel], vv[sel]), 1.0 - uu[sel] - vv[sel])
        return idx, tt, uu, vv, mm

    def any_hit(self, o, d, tmin, tmax, eps=0.0):
        """is the segment o + t d, tmin < t < tmax, blocked?  -> (bool, the smallest barycentric margin among the triangles whose answer decided it)"""
        a, b = o + d * tmin, o + d * tmax
        lo, hi = np.minimum(a, b) - 1e-9, np.maximum(a, b) + 1e-9
        cand = np.nonzero(((self.hi >= lo) & (self.lo <= hi)).all(1))[0]    # only triangles whose box the segment's box touches can be hit
        if cand.size == 0:
            return False, 1.0
        i, t, u, v, m = closest_hit(o, d, self.tris[cand], tmin, tmax, eps)
        return i >= 0, m


# ---- ray-traced ambient occlusion (libart's own definition behind XeGTAO's I/O contract: DESIGN.md 6 f2), restated in float64 ----------------------------
def hilbert_index_64(x, y):
    """XeGTAO.h:120-142 at XE_HILBERT_LEVEL 6: position of (x, y) on the Hilbert curve over a 64 x 64 tile"""
    index, lvl = 0, 32
    while lvl > 0:
        rx, ry = int((x & lvl) > 0), int((y & lvl) > 0)
        index += lvl * lvl * ((3 * rx) ^ ry)
        if ry == 0:
            if rx == 1:
                x, y = 63 - x, 63 - y
            x, y = y, x
        lvl //= 2
    return index


# the R2 sequence's two constants AS THE DEFINITION HAS THEM: rounded to float (the index reaches 8 400: the double constants would move a sample by 4e-4)
R2_A, R2_B = float(np.float32(0.75487766624669276)), float(np.float32(0.56984029099805327))


def ao_pixel(x, y, w, h, depth, normal_out, view_inv, proj_inv, tris, spp, radius, tol=1e-6):
    """the AO value (0..255) of one pixel from the frame's depth and view-space normal outputs: the position is the primary ray scaled to the view depth, the
    normal the normal output taken back to world space, the spp directions a cosine-weighted hemisphere about it -- the R2 sequence on the pixel's Hilbert index
    (index + 288 sample) through the concentric square-root map, in the frame of Duff et al. (2017) -- each an any-hit segment (0.01 r, r) against EVERY triangle;
    the count of blocked segments goes through (1 - k / spp) ^ 2.2 (XeGTAO's final power, vk_xe_gtao.rs:22)"""
    if not depth < 10000.0:
        return 255, 255, 255
    o, d = primary_ray(x, y, w, h, view_inv, proj_inv)
    pc = np.array([x + 0.5, y + 0.5]) / np.array([w, h], np.float64) * 2.0 - 1.0
    tn = normalize((proj_inv @ np.array([pc[0], pc[1], 1.0, 1.0]))[:3])
    wp = o + d * (depth / -tn[2])
    n = np.array([normal_out[0] * 2.0 - 1.0, -(normal_out[1] * 2.0 - 1.0), -(normal_out[2] * 2.0 - 1.0)])
    N = normalize(view_inv[:3, :3] @ n)
    sg = math.copysign(1.0, N[2])
    a = -1.0 / (sg + N[2]); b = N[0] * N[1] * a
    T = np.array([1.0 + sg * N[0] * N[0] * a, sg * b, -sg * N[0]])
    B = np.array([b, sg + N[1] * N[1] * a, -N[1]])
    hidx = hilbert_index_64(x & 63, y & 63)
    blocked, sure, maybe = 0, 0, 0
    for s in range(spp):
        fi = float(hidx + 288 * s)
        u1, u2 = math.modf(0.5 + fi * R2_A)[0], math.modf(0.5 + fi * R2_B)[0]
        r, cz, phi = math.sqrt(u1), math.sqrt(1.0 - u1), 2.0 * math.pi * u2
        dvec = T * (r * math.cos(phi)) + B * (r * math.sin(phi)) + N * cz
        blocked += closest_hit(wp, dvec, tris, radius * 0.01, radius)[0] >= 0
        # the same segment a hair shorter at both ends against triangles a hair smaller, and a hair longer against triangles a hair larger: a segment that ends ON a
        # surface or passes through an edge is blocked in one arithmetic and free in another
        sure += closest_hit(wp, dvec, tris, radius * 0.01 * (1 + tol), radius * (1 - tol), -tol)[0] >= 0
        maybe += closest_hit(wp, dvec, tris, radius * 0.01 * (1 - tol), radius * (1 + tol), tol)[0] >= 0
    val = lambda k: int(math.floor((1.0 - k / spp) ** 2.2 * 255.0 + 0.5))
    return val(blocked), val(maybe), val(sure)     # the value, and the interval it may lie in (fewest .. most blocked segments)


# ---- presentation: the B10G11R11 read-back, AO, AMD's LPM tone mapper as the reference configures it, display gamma -- restated in float64 ----------------
def unpack_ufloat(v, mantissa_bits):
    """an unsigned small float (5 exponent bits, bias 15) as the B10G11R11_UFLOAT_PACK32 format defines it"""
    e, m = v >> mantissa_bits, v & ((1 << mantissa_bits) - 1)
    if e == 31:
        return math.inf if m == 0 else math.nan
    if e == 0:
        return m * 2.0 ** (-14 - mantissa_bits)
    return (1.0 + m * 2.0 ** -mantissa_bits) * 2.0 ** (e - 15)


def lpm_setup_709():
    """LpmData::new(false, 0.0, 256.0, 8.0, 0.25, 1.0, zeros, (1, 1/2, 1/32)) with LPM_CONFIG_709_709 / LPM_COLORS_709_709 (vk_tonemap.rs:60-120, :417-426),
    get_control_block (:122-230) and its own LpmColXyToZ / LpmColRgbToXyz (:12-47, z = 1 - x + y as written there) -- the numbers LpmMap reads"""
    hdr_max, exposure, contrast, shoulder_contrast = 256.0, 8.0, 0.25 + 1.0, 1.0
    saturation = np.zeros(3) + contrast
    crosstalk = np.array([1.0, 1.0 / 2.0, 1.0 / 32.0])
    mid_in, mid_out = hdr_max * 0.18 * 2.0 ** -exposure, 0.18
    cs = contrast * shoulder_contrast
    z0 = -mid_in ** contrast
    z1 = hdr_max ** cs * mid_in ** contrast
    z2 = hdr_max ** contrast * mid_in ** cs * mid_out
    z3 = hdr_max ** cs * mid_out
    z4 = mid_in ** cs * mid_out
    tone = np.array([-((z0 + (mid_out * (z1 - z2)) / (z3 - z4)) / z4), (z1 - z2) / (z3 - z4)])

    def xy_to_z(s):
        return np.array([s[0], s[1], 1.0 - s[0] + s[1]])

    def rgb_to_xyz(r, g, b, w):
        rgb3 = np.stack([xy_to_z(r), xy_to_z(g), xy_to_z(b)], axis=1)       # columns r g b
        w3 = xy_to_z(w) / w[1]
        s = np.linalg.inv(rgb3) @ w3
        return rgb3 * s[None, :]                                             # every row scaled component-wise

    m = rgb_to_xyz((0.64, 0.33), (0.30, 0.60), (0.15, 0.06), (0.3127, 0.3290))
    luma_w = m[1] / m[1].sum()
    luma_t = m[1] / m[1].sum()                                               # soft = false: the working space's Y row
    return dict(saturation=saturation, contrast=contrast, tone=tone, luma_w=luma_w, luma_t=luma_t, rcp_luma_t=1.0 / luma_t, crosstalk=crosstalk)


def lpm_map_709(c, P):
    """LpmMap (ffx_lpm.h:727-832) with shoulder, con, soft, con2, clip and scaleOnly all false"""
    sat = lambda x: min(max(x, 0.0), 1.0)
    R, G, B = c
    rcp_max = 1.0 / max(R, G, B)
    ratio = np.array([R * rcp_max, G * rcp_max, B * rcp_max]) ** P["saturation"]
    lt = P["luma_t"]
    luma = G * lt[1] + (R * lt[0] + B * lt[2])
    luma = luma ** P["contrast"]
    luma = luma / (luma * P["tone"][0] + P["tone"][1])
    scale = sat(luma / float(ratio @ lt))
    col = np.array([sat(x * scale) for x in ratio])
    cap = -P["crosstalk"] * col + P["crosstalk"]
    add = sat(luma - float(col @ lt))
    t = add / float(cap @ lt)
    col = np.array([sat(t * cap[k] + col[k]) for k in range(3)])
    add = sat(luma - float(col @ lt))
    return np.array([sat(add * P["rcp_luma_t"][k] + col[k]) for k in range(3)])


def present_pixel(packed, ao, P):
    """tonemap.comp.glsl:32-38 on one stored pixel: the packed colour read back, times ao / 255, LpmFilter(..., LPM_CONFIG_709_709), pow(1 / 2.2) -> B, G, R as 8-bit UNORM"""
    c = np.array([unpack_ufloat(packed & 0x7FF, 6), unpack_ufloat((packed >> 11) & 0x7FF, 6), unpack_ufloat(packed >> 22, 5)]) * (ao / 255.0)
    c = lpm_map_709(c, P) if c.max() > 0.0 else np.zeros(3)
    c = np.clip(c ** (1.0 / 2.2), 0.0, 1.0) * 255.0
    return c[::-1]                                                               # B G R, before rounding
