"""CPU tests of the oracle itself (oracle/ is test infrastructure; PARITY UNPINNED -- the reference holds no golden
vectors for this path, so the oracle is pinned by analytic known answers, a brute-force cross-check, invariants, the
committed fixtures and an independent numpy restatement of the shading)."""
import json
import math
import os

import numpy as np
import pytest

import np_shading as NP
from conftest import assert_radiance_close
from helpers import oracle_camera, oracle_for, random_rays

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------------------------------------------ camera
def test_camera_block_closed_form(orc):
    cam = orc.camera_from_params((0, 0, 0), (0, 0, 1), 1.0, math.pi / 2, 0.1, 1000.0)   # renderer.rs:222-231 defaults
    view = np.array(cam.view, np.float64).reshape(4, 4).T
    # eye 0, dir +Z, up -Y  =>  view-x = world-x, view-y = -world-y, camera looks down world +Z
    assert np.allclose(view, np.diag([1, -1, -1, 1]), atol=1e-7)
    proj = np.array(cam.proj, np.float64).reshape(4, 4).T
    zn, zf = 0.1, 1000.0
    want = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, (zf + zn) / (zn - zf), 2 * zf * zn / (zn - zf)], [0, 0, -1, 0]])
    assert np.allclose(proj, want, rtol=1e-6)
    assert np.allclose(np.array(cam.view_inv).reshape(4, 4).T @ view, np.eye(4), atol=1e-6)
    assert np.allclose(np.array(cam.proj_inv).reshape(4, 4).T @ proj, np.eye(4), atol=1e-4)


def test_primary_rays_closed_form(orc):
    w = h = 256
    cam = orc.camera_from_params((0.5, -0.25, -0.95), (0, 0, 1), 1.0, math.pi / 2, 0.1, 1000.0)
    rays = orc.gen_primary(cam, w, h).reshape(h, w, 8)
    assert np.allclose(rays[..., 0:3], [0.5, -0.25, -0.95], atol=1e-6) and np.all(rays[..., 3] == np.float32(0.001)) and np.all(rays[..., 7] == 10000.0)
    for (x, y) in [(0, 0), (255, 0), (0, 255), (127, 128), (200, 13)]:
        dx, dy = (x + 0.5) / w * 2 - 1, (y + 0.5) / h * 2 - 1
        want = np.array([dx, -dy, 1.0])          # image row 0 is world +Y
        want /= np.linalg.norm(want)
        assert np.allclose(rays[y, x, 4:7], want, atol=2e-6), (x, y)


# ------------------------------------------------------------------------------------------------ ray / triangle
def _tri_scene(orc, scenes, tris):
    prims = []
    for t in tris:
        mb = scenes.MeshBuilder()
        mb.add(t, [(0, 0), (1, 0), (0, 1)], [(0, 0, -1)] * 3, [(1, 0, 0, 1)] * 3, [0, 1, 2])
        prims.append(mb.finish(scenes.constant_texture((200, 200, 200))))
    return orc.Scene(prims, morton_bits=30), prims


@pytest.mark.parametrize("mode", [0, 1])
def test_single_triangle_known_answers(orc, scenes, mode):
    S, _ = _tri_scene(orc, scenes, [[(0, 0, 2), (1, 0, 2), (0, 1, 2)]])
    rays = np.array([
        [0.25, 0.25, 0, 0.001, 0, 0, 1, 100],      # t=2, u=.25, v=.25
        [0.25, 0.25, 0, 0.001, 0, 0, -1, 100],     # behind
        [2.0, 2.0, 0, 0.001, 0, 0, 1, 100],        # outside
        [0.25, 0.25, 0, 0.001, 1, 0, 0, 100],      # parallel
        [0.25, 0.25, 0, 0.001, 0, 0, 1, 1.5],      # tmax too short
        [0.25, 0.25, 0, 2.5, 0, 0, 1, 100],        # tmin beyond
        [0.25, 0.25, 4, 0.001, 0, 0, -1, 100],     # back face: two-sided (instance flags 0, vk_model.rs:374)
        [0.5, 0.5, 0, 0.001, 0, 0, 1, 100],        # on the hypotenuse
        [0.25, 0.25, 0, 0.001, 0, 0, 1, 2.0],      # t == tmax: open interval, miss
        [0.25, 0.25, 0, 2.0, 0, 0, 1, 100],        # t == tmin: open interval, miss
    ], np.float32)
    tuv, ids, _, _ = S.trace_closest(rays, mode)
    assert ids[:, 0].tolist() == [0, -1, -1, -1, -1, -1, 0, 0, -1, -1]
    assert np.allclose(tuv[0, :3], [2, .25, .25]) and np.allclose(tuv[6, :3], [2, .25, .25]) and np.allclose(tuv[7, :3], [2, .5, .5])
    hit, _, _ = S.trace_any(rays, mode)
    assert hit.tolist() == [1, 0, 0, 0, 0, 0, 1, 1, 0, 0]


@pytest.mark.parametrize("mode", [0, 1])
def test_closest_of_two_and_tie_break(orc, scenes, mode):
    near, far = [(0, 0, 2), (1, 0, 2), (0, 1, 2)], [(0, 0, 3), (1, 0, 3), (0, 1, 3)]
    ray = np.array([[0.2, 0.2, 0, 0.001, 0, 0, 1, 100]], np.float32)
    for order, want in (([far, near], 1), ([near, far], 0)):
        S, _ = _tri_scene(orc, scenes, order)
        tuv, ids, _, _ = S.trace_closest(ray, mode)
        assert ids[0, 0] == want and tuv[0, 0] == 2.0
    S, _ = _tri_scene(orc, scenes, [near, near, near])      # coincident: the lowest global triangle id wins
    _, ids, _, _ = S.trace_closest(ray, mode)
    assert ids[0].tolist() == [0, 0]


def test_shared_edge_is_watertight(orc, scenes):
    """rays through the shared diagonal of a quad must not slip between its two triangles (hardware RT is watertight)"""
    mb = scenes.MeshBuilder()
    scenes.quad(mb, (-1, -1, 2), (2, 0, 0), (0, 2, 0))
    S = orc.Scene([mb.finish(scenes.constant_texture((1, 1, 1)))])
    n = 4001
    s = np.linspace(-0.999, 0.999, n)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0], rays[:, 1] = s, s                   # exactly on the diagonal
    rays[:, 3], rays[:, 6], rays[:, 7] = 0.001, 1.0, 100.0
    _, ids, _, _ = S.trace_closest(rays)
    assert (ids[:, 0] == 0).all()


# ------------------------------------------------------------------------------------------------ lights + BRDF
def test_light_records_and_layout(orc):
    p = orc.make_light(dict(kind="point", pos=(1, 2, 3), color=(4, 5, 6), falloff=7.0, casts_shadows=True))
    assert (p.type, p.casts_shadows, list(p.pos), list(p.color), p.falloff_distance) == (0, 1, [1, 2, 3], [4, 5, 6], 7.0)
    assert list(p.dir) == [0, 0, 0] and p.penumbra_angle == 0 and p.umbra_angle == 0
    s = orc.make_light(dict(kind="spot", pos=(0, 1.5, 0), dir=(0, -1, 0), color=(1, 1, 1), falloff=3.0, penumbra=0.5, umbra=0.8, casts_shadows=False))
    assert (s.type, s.casts_shadows, np.float32(s.penumbra_angle), np.float32(s.umbra_angle)) == (1, 0, np.float32(0.5), np.float32(0.8))
    d = orc.make_light(dict(kind="directional", dir=(0, -1, 0), color=(3, 3, 3), casts_shadows=True))
    assert d.type == 2 and d.falloff_distance == 0.0
    a = orc.make_light(dict(kind="area", pos=(-0.7, 0.77, 0.08), pos2=(-0.7, 0.77, -0.16), pos3=(-0.7, 0.9, -0.16), invert_normal=False, color=(1, 1, 1),
                            falloff=3.0, penumbra=1.0, umbra=1.2, casts_shadows=True))
    # lights.rs:385-389: normalize((pos - pos2) x (pos3 - pos2)) = normalize((0,0,.24) x (0,.13,0)) = (-1,0,0)
    assert a.type == 3 and np.allclose(list(a.dir), [-1, 0, 0], atol=1e-6)
    ai = orc.make_light(dict(kind="area", pos=(-0.7, 0.77, 0.08), pos2=(-0.7, 0.77, -0.16), pos3=(-0.7, 0.9, -0.16), invert_normal=True, color=(1, 1, 1),
                             falloff=3.0, penumbra=1.0, umbra=1.2, casts_shadows=True))
    assert np.allclose(list(ai.dir), [1, 0, 0], atol=1e-6)


def test_light_vectors_and_radiance_known_answers(orc):
    pt = orc.make_light(dict(kind="point", pos=(0, 2, 0), color=(8, 8, 8), falloff=4.0, casts_shadows=True))
    nn, rad = orc.light_eval(pt, (0, 0, 0))
    assert np.allclose(nn, [0, 2, 0]) and np.allclose(rad, 8 * (1 - 0.25) ** 2)           # max(1-(d/f)^2,0)^2
    _, rad = orc.light_eval(pt, (0, -3, 0))
    assert np.allclose(rad, 0)                                                                # beyond the falloff distance
    dl = orc.make_light(dict(kind="directional", dir=(0, -1, 0), color=(3, 3, 3), casts_shadows=True))
    nn, rad = orc.light_eval(dl, (5, 5, 5))
    assert np.allclose(nn, [0, 10, 0]) and np.allclose(rad, 3)                               # light.glsl:97-99: -dir * 10
    sp = orc.make_light(dict(kind="spot", pos=(0, 1, 0), dir=(0, -1, 0), color=(1, 1, 1), falloff=0.0, penumbra=math.radians(30), umbra=math.radians(45), casts_shadows=True))
    _, rad = orc.light_eval(sp, (0, 0, 0))                                                   # on axis: theta=0 -> t = (0-u)/(p-u) = 3 -> clamp 1
    assert np.allclose(rad, 1)
    ang = math.radians(37.5)
    _, rad = orc.light_eval(sp, (math.tan(ang), 0, 0))                                       # half way between penumbra and umbra: t = .5
    assert np.allclose(rad, 0.25, rtol=1e-4)
    _, rad = orc.light_eval(sp, (math.tan(math.radians(50)), 0, 0))
    assert np.allclose(rad, 0)
    # area light: parallelogram pos=(0,1,0) pos2=(1,1,0) pos3=(1,1,1) => pos4=(0,1,1); plane y=1
    ar = dict(kind="area", pos=(0, 1, 0), pos2=(1, 1, 0), pos3=(1, 1, 1), invert_normal=False, color=(1, 1, 1), falloff=0.0, penumbra=math.radians(90),
              umbra=math.radians(90), casts_shadows=True)
    al = orc.make_light(ar)
    for p, want in [((0.6, 0, 0.3), (0.6, 1, 0.3)),      # inside triangle pos,pos2,pos3
                    ((0.3, 0, 0.6), (0.3, 1, 0.6)),      # inside the other half (pos, pos3, pos4)
                    ((0.5, 0, -1.0), (0.5, 1, 0.0)),     # beyond edge pos-pos2
                    ((2.0, 0, 0.5), (1.0, 1, 0.5)),      # beyond edge pos2-pos3
                    ((0.5, 0, 3.0), (0.5, 1, 1.0)),      # beyond edge pos3-pos4
                    ((-2.0, 0, 0.5), (0.0, 1, 0.5))]:    # beyond edge pos4-pos
        nn, _ = orc.light_eval(al, p)
        ref = NP.get_unnormalized_L_vec(NP.light_from_record(al), np.array(p, np.float64))
        assert np.allclose(nn, ref, atol=1e-6), (p, nn, ref)
        assert np.allclose(nn, np.array(want) - np.array(p), atol=1e-5), (p, nn)


def test_brdf_terms_against_closed_forms(orc):
    for (NdotL, NdotV, NdotH, LdotH, alpha) in [(0.7, 0.5, 0.9, 0.8, 0.25), (0.2, 0.9, 0.4, 0.3, 0.81), (1.0, 1e-5, 1.0, 1.0, 0.04), (0.05, 0.3, 0.999, 0.1, 0.5)]:
        got = orc.brdf_terms(NdotL, NdotV, NdotH, LdotH, NdotV, NdotL, alpha)
        want = [NP.D_GGX(alpha, NdotH), NP.V_SmithGGXCorrelated_fast(alpha, NdotV, NdotL), (1 - LdotH) ** 5, NP.Burley_diffuse_local_sss(alpha, NdotV, NdotV, NdotL, LdotH, 0.4)]
        assert np.allclose(got, want, rtol=2e-5), (got, want)


# ------------------------------------------------------------------------------------------------ whole pixels
def _floor_scene(scenes, blocker_height=None):
    prims = []
    mb = scenes.MeshBuilder()
    scenes.quad(mb, (-0.9, 0, -0.9), (0, 0, 1.8), (1.8, 0, 0))   # floor y=0, normal +y
    prims.append(mb.finish(scenes.constant_texture((200, 200, 200))))
    if blocker_height is not None:
        mb = scenes.MeshBuilder()
        scenes.quad(mb, (-0.9, blocker_height, -0.9), (0, 0, 1.8), (1.8, 0, 0))
        prims.append(mb.finish(scenes.constant_texture((200, 200, 200))))
    cam = dict(pos=(0.0, 0.5, 0.0), dir=(0.0, -1.0, 0.001), fovy=math.pi / 3, znear=0.1, zfar=1000.0)
    return prims, cam


def _centre(orc, prims, cam, lights):
    sc_lights = orc.make_lights(lights)
    S = orc.Scene(prims)
    c = orc.camera_from_params(cam["pos"], cam["dir"], 1.0, cam["fovy"], cam["znear"], cam["zfar"])
    out = S.render(c, sc_lights, len(lights), 8, 8, debug=True)
    return out


def test_shadowed_light_keeps_five_percent_and_directional_range_is_ten(orc, scenes):
    """raytrace.rgen.glsl:179-181 (0.05) and light.glsl:97-99 (tmax = |-dir*10| = 10 for directional lights)"""
    light = [dict(kind="directional", dir=(0, -1, 0), color=(3, 3, 3), casts_shadows=True)]
    prims, cam = _floor_scene(scenes)
    lit = _centre(orc, prims, cam, light)
    # the camera (y=.5) looks down; a blocker above the camera is not seen by primary rays but is crossed by shadow rays
    for height, shadowed in ((9.0, True), (11.0, False)):
        prims_b, _ = _floor_scene(scenes, blocker_height=height)
        out = _centre(orc, prims_b, cam, light)
        sb = out["shadow_bits"][4, 4]
        assert bool(sb & 1) == shadowed and bool(sb >> 16 & 1)
        ratio = out["color"][4, 4, :3] / lit["color"][4, 4, :3]
        assert np.allclose(ratio, 0.05 if shadowed else 1.0, rtol=1e-5)
    assert lit["stats"]["shadow_rays"] == 64 and lit["stats"]["hit_pixels"] == 64


def test_miss_outputs(orc, scenes):
    prims, cam = _floor_scene(scenes)
    cam = dict(cam, dir=(0.0, 1.0, 0.001))        # look away from the floor
    out = _centre(orc, prims, cam, [dict(kind="directional", dir=(0, -1, 0), color=(3, 3, 3), casts_shadows=True)])
    assert (out["hit_id"] == -1).all()
    assert np.array_equal(out["color"][..., :3], np.zeros((8, 8, 3), np.float32)) and (out["color"][..., 3] == 1).all()   # raytrace.rgen.glsl:103-105,197
    assert (out["depth"] == 10000.0).all() and np.array_equal(out["normal"][..., :3], np.full((8, 8, 3), 0.5, np.float32))
    assert out["stats"]["shadow_rays"] == 0


def test_shading_matches_independent_numpy_restatement(orc, scenes):
    """every light type, normal-mapped + textured surfaces: C oracle vs the numpy restatement written from the GLSL"""
    sc = scenes.sponza_like(0.05)
    lights = scenes.sponza_lights(4)
    w, h = 48, 27
    S = orc.Scene(sc.primitives)
    cam = oracle_camera(orc, sc, w, h)
    recs = orc.make_lights(lights)
    out = S.render(cam, recs, 4, w, h, debug=True)
    view = np.array(cam.view, np.float64).reshape(4, 4).T
    view_inv = np.array(cam.view_inv, np.float64).reshape(4, 4).T
    nl = [NP.light_from_record(recs[i]) for i in range(4)]
    checked = 0
    for y in range(0, h, 2):
        for x in range(0, w, 3):
            pi, ti = out["hit_id"][y, x]
            if pi < 0:
                continue
            _, u, v, _ = out["hit_tuv"][y, x]
            rho, depth, on, mask = NP.shade_pixel(sc.primitives[pi], int(ti), float(u), float(v), view, view_inv, np.array(cam.camera_pos, np.float64), nl,
                                                  int(out["shadow_bits"][y, x]) & 0xFFFF)
            assert mask == int(out["shadow_bits"][y, x]) & 0xFFFF0000, (x, y)
            assert np.allclose(out["color"][y, x, :3], rho, rtol=2e-4, atol=2e-6), (x, y, out["color"][y, x], rho)
            assert np.isclose(out["depth"][y, x], depth, rtol=1e-5) and np.allclose(out["normal"][y, x, :3], on, atol=1e-5)
            checked += 1
    assert checked > 80


# ---- the independent leg, tightened: art_oracle.c compiled with every float a double (liborc_f64.so, oracle/orc64.py) against
# the numpy restatement, BOTH in fp64, so that what is left between them is transcription, not rounding; then the float build against the
# double build of the same source, which is rounding only.
@pytest.fixture(scope="module")
def orc64():
    from oracle import orc64 as o
    o.build()
    return o


def _f64_frame(orc64, sc, lights, w, h):
    S = orc64.Scene(sc.primitives)
    cam = oracle_camera(orc64, sc, w, h)
    recs = orc64.make_lights(lights)
    return S, cam, recs, S.render(cam, recs, len(lights), w, h, threads=4, debug=True)


def test_cornell_gbuffer_geometry_and_shading_against_numpy_in_fp64(orc64, scenes):
    """the whole Cornell 64x64 G-buffer: camera block and primary rays against the closed forms, every pixel's hit against a brute-force fp64
    Moeller-Trumbore over the 34 triangles (id, t, u, v), every pixel's colour / depth / normal against the numpy shading: 1e-6 relative
    (the constants of the C source keep their float-rounded values, ~3e-8; everything else agrees to ~1e-12)"""
    sc = scenes.cornell()
    w = h = 64
    S, cam, recs, out = _f64_frame(orc64, sc, sc.lights, w, h)
    c = sc.camera
    view, view_inv, proj, proj_inv = NP.camera_matrices(c["pos"], c["dir"], w / h, c["fovy"], c["znear"], c["zfar"])
    for name, want in (("view", view), ("view_inv", view_inv), ("proj", proj), ("proj_inv", proj_inv)):
        assert np.allclose(np.array(getattr(cam, name)).reshape(4, 4).T, want, rtol=1e-12, atol=1e-12), name
    rays = orc64.gen_primary(cam, w, h).reshape(h, w, 8)
    tris, pid, tid = NP.world_triangles(sc.primitives)
    lights = [NP.light_from_record(recs[i]) for i in range(len(sc.lights))]
    cam_pos = np.array(cam.camera_pos, np.float64)
    edge_cases = 0
    for y in range(h):
        for x in range(w):
            o, d = NP.primary_ray(x, y, w, h, view_inv, proj_inv)
            assert np.allclose(rays[y, x, 0:3], o, atol=1e-12) and np.allclose(rays[y, x, 4:7], d, atol=1e-12) and rays[y, x, 3] == np.float32(0.001) and rays[y, x, 7] == 10000.0
            i, t, u, v, margin = NP.closest_hit(o, d, tris, float(np.float32(0.001)))
            pi, ti = out["hit_id"][y, x]
            if i < 0 or (pid[i], tid[i]) != (pi, ti):
                # only where the ray passes within the oracle's edge tolerance (1e-6 barycentric) of a shared edge: plain Moeller-Trumbore can miss
                # BOTH triangles there (the frame's diagonal crosses the back wall's); with the edges fattened alike it finds one of the two
                edge_cases += 1
                i2, t2, u2, v2, m2 = NP.closest_hit(o, d, tris, float(np.float32(0.001)), eps=1e-6)
                assert i2 >= 0 and pi >= 0 and m2 < 2e-6 and np.isclose(out["hit_tuv"][y, x, 0], t2, rtol=1e-9), (x, y, i, i2, pi, ti, margin, m2)
                continue
            assert np.allclose(out["hit_tuv"][y, x, :3], [t, u, v], rtol=1e-9, atol=1e-11), (x, y)
            rho, depth, on, mask = NP.shade_pixel(sc.primitives[pi], int(ti), u, v, view, view_inv, cam_pos, lights, int(out["shadow_bits"][y, x]) & 0xFFFF)
            assert mask == int(out["shadow_bits"][y, x]) & 0xFFFF0000
            assert np.allclose(out["color"][y, x, :3], rho, rtol=1e-6, atol=1e-9), (x, y, out["color"][y, x], rho)
            assert np.isclose(out["depth"][y, x], depth, rtol=1e-9) and np.allclose(out["normal"][y, x, :3], on, rtol=1e-6, atol=2e-7)
    assert edge_cases <= 32 and out["stats"]["hit_pixels"] == w * h   # (the frame's diagonal runs along the back wall's: a few of its 64 pixels land on the shared edge)


def test_all_light_types_against_numpy_in_fp64(orc64, scenes):
    """point + spot + directional + area light on normal-mapped, textured surfaces (the sample of test_shading_matches_independent_numpy_restatement,
    four times denser): 1e-6 relative + 2e-7 in fp64, where the float build needed 2e-4"""
    sc = scenes.sponza_like(0.05)
    lights = scenes.sponza_lights(4)
    w, h = 96, 54
    S, cam, recs, out = _f64_frame(orc64, sc, lights, w, h)
    view = np.array(cam.view, np.float64).reshape(4, 4).T
    view_inv = np.array(cam.view_inv, np.float64).reshape(4, 4).T
    nl = [NP.light_from_record(recs[i]) for i in range(4)]
    checked, lit = 0, np.zeros(4, int)
    for y in range(0, h, 2):
        for x in range(0, w, 2):
            pi, ti = out["hit_id"][y, x]
            if pi < 0:
                continue
            _, u, v, _ = out["hit_tuv"][y, x]
            rho, depth, on, mask = NP.shade_pixel(sc.primitives[pi], int(ti), float(u), float(v), view, view_inv, np.array(cam.camera_pos, np.float64), nl,
                                                  int(out["shadow_bits"][y, x]) & 0xFFFF)
            assert mask == int(out["shadow_bits"][y, x]) & 0xFFFF0000, (x, y)
            # (2e-7 absolute: where a falloff or cone window closes, 1 - q^2 amplifies the ~3e-8 by which the C source's float-rounded constants differ from numpy's)
            assert np.allclose(out["color"][y, x, :3], rho, rtol=1e-6, atol=2e-7), (x, y, out["color"][y, x], rho)
            assert np.isclose(out["depth"][y, x], depth, rtol=1e-9) and np.allclose(out["normal"][y, x, :3], on, rtol=1e-6, atol=2e-7)
            checked += 1
            lit += [(mask >> (16 + i)) & 1 for i in range(4)]
    assert checked > 1000 and (lit > 20).all(), (checked, lit)


def test_repeat_sampler_at_the_texture_seams_against_numpy_in_fp64(orc64, scenes):
    """the last piece of the shading both hands wrote alike: bilinear filtering with REPEAT where the 2x2 footprint wraps -- uv below 0, at 1, past 1; the
    texel before column 0 is the last column.  One quad whose uv run from -1.25 to 2.25 (and -0.75 to 1.75) under a 4 x 5 texture of unrelated texels, so nearly every pixel's
    footprint sits on a texel boundary and a fifth of them on the wrap: the fp64 oracle against the numpy sampler (Python's floor-mod, no shared helper) through the whole shading
    of every pixel -- albedo, roughness / metallic and the normal map all go through it -- at 5e-6 (normal: 1e-6)"""
    from helpers import seam_scene
    sc = seam_scene(scenes)
    prim = sc.primitives[0]
    th, tw = prim.tex.shape[1:3]
    w = h = 96
    S, cam, recs, out = _f64_frame(orc64, sc, sc.lights, w, h)
    view = np.array(cam.view, np.float64).reshape(4, 4).T
    view_inv = np.array(cam.view_inv, np.float64).reshape(4, 4).T
    nl = [NP.light_from_record(recs[0])]
    V = prim.verts.astype(np.float64)
    checked = wrapped = negative = 0
    for y in range(h):
        for x in range(w):
            pi, ti = out["hit_id"][y, x]
            if pi < 0:
                continue
            _, u, v, _ = out["hit_tuv"][y, x]
            i0, i1, i2 = [int(k) for k in prim.indices[3 * ti:3 * ti + 3]]
            uv = V[i0, 3:5] * (1 - u - v) + V[i1, 3:5] * u + V[i2, 3:5] * v
            x0, y0 = math.floor(uv[0] * tw - 0.5), math.floor(uv[1] * th - 0.5)
            wrapped += (x0 % tw == tw - 1) or (y0 % th == th - 1)                     # the footprint's second column / row is the first one again
            negative += x0 < 0 or y0 < 0
            rho, depth, on, mask = NP.shade_pixel(prim, int(ti), float(u), float(v), view, view_inv, np.array(cam.camera_pos, np.float64), nl, 0)
            assert np.allclose(out["color"][y, x, :3], rho, rtol=5e-6, atol=2e-7), (x, y, uv, out["color"][y, x], rho)   # (a wrong texel shows at 1e-2; the C source's float-rounded constants at 1e-6 under this bright a light)
            assert np.allclose(out["normal"][y, x, :3], on, rtol=1e-6, atol=2e-7), (x, y, uv)
            checked += 1
    assert checked > 3000 and wrapped > checked // 6 and negative > checked // 8, (checked, wrapped, negative)


@pytest.mark.parametrize("config,n_pixels", [("c2", 2000), ("c4", 200)])
def test_bench_scenes_against_a_brute_force_witness_in_fp64(orc, orc64, scenes, config, n_pixels):
    """The scenes the bench runs, at FULL detail -- config 2 (262 816 triangles, a primitive with u32 indices) and config 4 (2.8 M triangles, 63-bit Morton keys on
    the GPU) -- where until round 3 geometry was only checked oracle-BVH against oracle-brute-force: for a random sample of the 1080p frame's pixels, numpy in
    float64 tests the camera ray against EVERY triangle (no tree, no keys, no traversal order: tests/np_shading.py BruteForce), and
      - the hit's primitive, triangle, t, u, v are the fp64 build of the oracle's (its BVH walk), and the float oracle's ids -- the checker of every GPU test;
      - the pixel is shaded by the numpy restatement FROM NUMPY'S OWN HIT, its shadow ray traced by brute force too: colour, depth, normal against the oracle's.
    A transcription slip that only shows on large index ranges (u32 indices, global triangle ids past 2^16, primitive tables of 120 entries) would have to be made
    twice, in two languages."""
    import np_shading as NP
    sc = scenes.sponza_like(1.0) if config == "c2" else scenes.bistro_like(1.0)
    lights = scenes.sponza_lights(1) if config == "c2" else sc.lights
    w, h = 1920, 1080
    assert sc.n_tris > (250_000 if config == "c2" else 2_500_000) and (config != "c2" or any(p.indices.dtype == np.uint32 for p in sc.primitives))   # config 2 holds a primitive with u32 indices
    S, cam, recs, out = _f64_frame(orc64, sc, lights, w, h)
    c = sc.camera
    view, view_inv, proj, proj_inv = NP.camera_matrices(c["pos"], c["dir"], w / h, c["fovy"], c["znear"], c["zfar"])
    rng = np.random.default_rng(0xA17 + len(config))
    xs, ys = rng.integers(0, w, n_pixels), rng.integers(0, h, n_pixels)
    rays = np.array([np.concatenate(NP.primary_ray(int(x), int(y), w, h, view_inv, proj_inv)) for x, y in zip(xs, ys)])
    o, D = rays[0, :3], rays[:, 3:]
    assert np.allclose(rays[:, :3], o)
    B = NP.BruteForce(sc.primitives)
    tmin = float(np.float32(0.001))
    idx, t, u, v, margin = B.closest_from(o, D, tmin, 10000.0)
    S32 = orc.Scene(sc.primitives, morton_bits=63 if config == "c4" else 30)
    r32 = np.zeros((n_pixels, 8), np.float32); r32[:, :3] = o; r32[:, 3] = 0.001; r32[:, 4:7] = D; r32[:, 7] = 10000.0
    _, ids32, _, _ = S32.trace_closest(r32)
    nl = [NP.light_from_record(recs[i]) for i in range(len(lights))]
    cam_pos = np.array(cam.camera_pos, np.float64)
    edge, hits, lit, shadowed, shadow_edge, prims_seen = 0, 0, 0, 0, 0, set()
    for k in range(n_pixels):
        x, y = int(xs[k]), int(ys[k])
        pi, ti = out["hit_id"][y, x]
        mine = (int(B.pid[idx[k]]), int(B.tid[idx[k]])) if idx[k] >= 0 else (-1, -1)
        if mine != (pi, ti):   # only within the oracle's edge tolerance (1e-6 barycentric) of an edge may the two name different triangles (or one a miss)
            i2, t2, u2, v2, m2 = NP.closest_hit(o, D[k], B.tris, tmin, eps=1e-6)
            assert (margin[k] < 2e-6 or m2 < 2e-6) and (pi < 0 or np.isclose(out["hit_tuv"][y, x, 0], t2, rtol=1e-7)), (x, y, mine, (pi, ti), margin[k], m2)
            edge += 1
            continue
        assert tuple(ids32[k]) == mine or margin[k] < 1e-4, (x, y, mine, tuple(ids32[k]), margin[k])     # the float oracle: the same triangle unless the ray grazes an edge
        if pi < 0:
            continue
        hits += 1; prims_seen.add(int(pi))
        assert np.allclose(out["hit_tuv"][y, x, :3], [t[k], u[k], v[k]], rtol=1e-9, atol=1e-11), (x, y)
        grazing = []

        def occluded(i, origin, L, tmax):
            hit, m = B.any_hit(origin, L, float(np.float32(0.01)), tmax, eps=1e-6)
            grazing.append(m < 2e-6)
            return hit
        rho, depth, on, mask = NP.shade_pixel(sc.primitives[pi], int(ti), u[k], v[k], view, view_inv, cam_pos, nl, occluded)
        bits = int(out["shadow_bits"][y, x])
        assert mask & 0xFFFF0000 == bits & 0xFFFF0000, (x, y)
        lit += (mask >> 16) & 1
        if mask & 0xFFFF != bits & 0xFFFF:      # the shadow ray grazes an edge of the blocker (or the blocker lies within rounding of the ray's start): either answer
            shadow_edge += 1
            continue
        shadowed += mask & 1
        assert np.allclose(out["color"][y, x, :3], rho, rtol=1e-6, atol=2e-7), (x, y, out["color"][y, x], rho)
        assert np.isclose(out["depth"][y, x], depth, rtol=1e-9) and np.allclose(out["normal"][y, x, :3], on, rtol=1e-6, atol=2e-7)
    assert hits > 0.6 * n_pixels and edge <= n_pixels // 100 and shadow_edge <= max(2, lit // 50), (hits, edge, shadow_edge, lit)
    assert lit > hits // 10 and 0 < shadowed < lit and len(prims_seen) >= 8, (lit, shadowed, prims_seen)


def test_float_build_is_the_double_build_up_to_rounding(orc, orc64, scenes):
    """the oracle everything else is compared with (float) against the double build of the same source.  On the Cornell G-buffer (constant
    textures) the radiance agrees to 1e-5 everywhere: SURVEY.md 8c-5's figure.  On textured, normal-mapped surfaces with all four light
    types the float build's own rounding shows: the hit point moves by ~1e-7 and the texture / normal-map gradients, the closing falloff
    and cone windows and grazing N.V * N.L amplify that -- median 2e-6, a tail to ~1e-3.  (That is the float PIPELINE against exact
    arithmetic, not GPU against oracle: those two share every geometry bit, and differ by 5.6e-6 at worst, tests/golden/*.stats.json.)"""
    sc = scenes.cornell()
    w = h = 64
    a = oracle_for(orc, sc)[0].render(oracle_camera(orc, sc, w, h), orc.make_lights(sc.lights), 1, w, h, threads=4, debug=True)
    b = _f64_frame(orc64, sc, sc.lights, w, h)[3]
    assert np.array_equal(a["hit_id"], b["hit_id"]) and np.array_equal(a["shadow_bits"], b["shadow_bits"])
    assert np.allclose(a["color"], b["color"], rtol=1e-5, atol=1e-9) and np.allclose(a["depth"], b["depth"], rtol=1e-6) and np.allclose(a["normal"], b["normal"], atol=1e-6)
    sc = scenes.sponza_like(0.05)
    lights = scenes.sponza_lights(4)
    w, h = 192, 108
    a = oracle_for(orc, sc)[0].render(oracle_camera(orc, sc, w, h), orc.make_lights(lights), 4, w, h, threads=4, debug=True)
    b = _f64_frame(orc64, sc, lights, w, h)[3]
    same = (a["hit_id"] == b["hit_id"]).all(-1) & (a["shadow_bits"] == b["shadow_bits"])
    assert same.mean() > 0.995                           # a ray through an edge or a shadow boundary may fall either way
    ca, cb = a["color"][same][:, :3].astype(np.float64), b["color"][same][:, :3]
    rel = (np.abs(ca - cb) / (np.abs(cb) + 1e-6)).max(1)
    # (no bound on the maximum: the main.rs:55-64 area light has penumbra == umbra, a step in theta -- a pixel on it takes either side)
    assert np.median(rel) < 5e-6 and np.percentile(rel, 90) < 1e-4 and np.percentile(rel, 99) < 1e-3, (np.median(rel), np.percentile(rel, [90, 99]), rel.max())
    assert np.allclose(a["depth"][same], b["depth"][same], rtol=1e-5) and np.percentile(np.abs(a["normal"][same] - b["normal"][same]), 99) < 1e-4


# ------------------------------------------------------------------------------------------------ LBVH + traversal
@pytest.mark.parametrize("name,detail,bits", [("cornell", 1.0, 30), ("sponza_like", 0.05, 30), ("sponza_like", 0.05, 63)])
def test_lbvh_invariants(orc, get_scene, name, detail, bits):
    sc = get_scene(name, detail)
    S = orc.Scene(sc.primitives, morton_bits=bits)
    b = S.lbvh()
    T = S.n_tris
    assert T == sc.n_tris
    keys, gid = b["keys"].astype(object), b["leaf_gid"].astype(object)
    comp = [k * (1 << 32) + g for k, g in zip(keys, gid)]
    assert all(comp[i] < comp[i + 1] for i in range(T - 1))               # (key, gid) strictly increasing
    assert sorted(b["leaf_gid"].tolist()) == list(range(T))
    child = b["child"]
    seen_leaf, seen_int = np.zeros(T, int), np.zeros(T - 1, int)
    lo, hi = np.concatenate([b["node_lo"], b["leaf_lo"]]), np.concatenate([b["node_hi"], b["leaf_hi"]])
    idx = lambda c: (T - 1 + (~c)) if c < 0 else c
    for n in range(T - 1):
        for c in child[n]:
            if c < 0:
                seen_leaf[~c] += 1
            else:
                seen_int[c] += 1
        a, bb = idx(child[n, 0]), idx(child[n, 1])
        assert np.array_equal(lo[n], np.minimum(lo[a], lo[bb])) and np.array_equal(hi[n], np.maximum(hi[a], hi[bb]))   # exact union
    assert (seen_leaf == 1).all() and seen_int[0] == 0 and (seen_int[1:] == 1).all()
    tv = b["tri_verts"].reshape(T, 3, 3)[b["leaf_gid"]]
    assert np.array_equal(b["leaf_lo"], tv.min(1)) and np.array_equal(b["leaf_hi"], tv.max(1))


@pytest.mark.parametrize("name,detail", [("cornell", 1.0), ("sponza_like", 0.03)])
def test_bvh_traversal_equals_brute_force(orc, get_scene, name, detail):
    """the structure-independent definition: argmin over ALL triangles of (t_eff, gid) -- exact equality"""
    sc = get_scene(name, detail)
    rays = random_rays(10000, 3)
    S30, S63 = orc.Scene(sc.primitives, morton_bits=30), orc.Scene(sc.primitives, morton_bits=63)
    tb, ib, _, _ = S30.trace_closest(rays, 1)
    for S in (S30, S63):
        t, i, n_int, n_tri = S.trace_closest(rays, 0)
        assert np.array_equal(i, ib) and np.array_equal(t.view(np.uint32), tb.view(np.uint32))
        assert n_int > 0 and n_tri < 10000 * sc.n_tris
    short = rays.copy()
    short[:, 7] = 1.2
    hb, _, _ = S30.trace_any(short, 1)
    for S in (S30, S63):
        hh, _, _ = S.trace_any(short, 0)
        assert np.array_equal(hh, hb)
    assert 0 < hb.sum() < hb.size


# ------------------------------------------------------------------------------------------------ fixtures
@pytest.mark.parametrize("n", [64, 256])
def test_cornell_golden_fixture(orc, get_scene, n):
    sc = get_scene("cornell")
    S, L, nl = oracle_for(orc, sc)
    out = S.render(oracle_camera(orc, sc, n, n), L, nl, n, n, threads=4)
    gold = np.load(os.path.join(GOLD, f"cornell_{n}.npz"))
    assert_radiance_close(out["color"], gold["color"], rel=1e-5)
    assert_radiance_close(out["depth"], gold["depth"], rel=1e-6, what="depth")
    assert_radiance_close(out["normal"], gold["normal"], rel=1e-5, what="normal")
    gold_st = json.load(open(os.path.join(GOLD, f"cornell_{n}.stats.json")))
    assert out["stats"] == {k: v for k, v in gold_st.items() if not k.startswith("packet_")}   # (the packet counts: test_packet_visit_counts)
    assert out["stats"]["nonfinite_pixels"] == 0 and sc.n_tris == 34


def test_threads_do_not_change_the_frame(orc, get_scene):
    sc = get_scene("cornell")
    S, L, nl = oracle_for(orc, sc)
    a = S.render(oracle_camera(orc, sc, 96, 64), L, nl, 96, 64, threads=1)
    b = S.render(oracle_camera(orc, sc, 96, 64), L, nl, 96, 64, threads=5)
    assert np.array_equal(a["color"], b["color"]) and a["stats"] == b["stats"]


def test_packet_visit_counts(orc, get_scene):
    """orc_packet_stats (bench.py's packet-level roofline figure): with 1x1 blocks a packet is one ray, so the counts ARE the per-ray
    counters; with 8x8 blocks every node / triangle counts once per block, so they lie between per-ray / 64 and per-ray; the frame's
    per-ray stats do not depend on how it is cut; committed for the Cornell fixture"""
    sc = get_scene("cornell")
    S, L, nl = oracle_for(orc, sc)
    cam = oracle_camera(orc, sc, 64, 64)
    ref = S.render(cam, L, nl, 64, 64, threads=3)["stats"]
    one, st1 = orc.packet_stats(S, cam, L, nl, 64, 64, block=(1, 1), threads=3)
    assert st1 == ref
    assert (one["packet_nodes_primary"], one["packet_tris_primary"], one["packet_nodes_shadow"], one["packet_tris_shadow"]) == \
           (ref["n_int_primary"], ref["n_tri_primary"], ref["n_int_shadow"], ref["n_tri_shadow"])
    pk, st8 = orc.packet_stats(S, cam, L, nl, 64, 64, threads=3)
    assert st8 == ref and orc.packet_stats(S, cam, L, nl, 64, 64, threads=1)[0] == pk
    for a, b in (("packet_nodes_primary", "n_int_primary"), ("packet_tris_primary", "n_tri_primary"), ("packet_nodes_shadow", "n_int_shadow"), ("packet_tris_shadow", "n_tri_shadow")):
        assert ref[b] / 64 <= pk[a] < ref[b], (a, pk[a], ref[b])
    assert pk["packet_nodes_primary"] >= 64                                   # 64 blocks, each visits at least the root
    gold = json.load(open(os.path.join(GOLD, "cornell_64.stats.json")))
    assert dict(ref, **pk) == gold
    whole, _ = orc.packet_stats(S, cam, L, nl, 64, 64, block=(64, 64))         # the whole frame as one packet: no node twice
    assert whole["packet_nodes_primary"] <= sc.n_tris - 1 and whole["packet_tris_primary"] <= sc.n_tris


def test_oracle_is_clean_under_the_sanitizers(tmp_path):
    """oracle/Makefile's asan target (AddressSanitizer + UBSan build of art_oracle.c), run on the paths every other test leans on: scene build,
    frame, packet counts, AO, ray queries, brute force -- in a child process with libasan preloaded; any report fails it"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(libasan), "gcc has no libasan.so"
    code = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from araytracingjourney_amd import scenes
from oracle import orc
from helpers import oracle_camera, random_rays
orc.build()
for sc, (w, h), lights in ((scenes.cornell(), (48, 40), None), (scenes.sponza_like(0.03), (64, 36), scenes.sponza_lights(4))):
    lights = sc.lights if lights is None else lights
    for bits in (30, 63):
        S = orc.Scene(sc.primitives, morton_bits=bits)
        cam = oracle_camera(orc, sc, w, h)
        L = orc.make_lights(lights)
        out = S.render(cam, L, len(lights), w, h, threads=3, debug=True)
        pk, st = orc.packet_stats(S, cam, L, len(lights), w, h, threads=3)
        assert st == out["stats"]
        ao, _ = orc.render_ao(S, cam, out["depth"], out["normal"], 5, 0.29, threads=3)
        rays = random_rays(500, 5)
        a = S.trace_closest(rays, 0); b = S.trace_closest(rays, 1)
        assert np.array_equal(a[1], b[1])
        S.trace_any(rays, 0); S.trace_any(rays, 1); S.lbvh()
        orc.present(out["color"], ao)
print("SANITIZER_RUN_OK")
""" % (root, os.path.join(root, "tests"))
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", ORC_SO="liborc_asan.so")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "SANITIZER_RUN_OK" in out.stdout and "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stdout[-1500:] + out.stderr[-3000:]


def test_ray_traced_ao_against_numpy_in_fp64(orc64, scenes):
    """the AO pass is libart's own definition (no reference parity possible: it stands in for XeGTAO behind the same inputs and output), so until round 4 its oracle
    was checked by the kernel written from it and nothing else.  The independent leg: the fp64 build of the oracle against tests/np_shading.py ao_pixel -- Hilbert
    index, R2 samples (with the two constants rounded to float, as the definition has them: with the double constants five pixels of 1 600 differed by one sample --
    the index reaches 8 400 and moves a direction by 4e-4), the hemisphere frame, numpy's own cos / sin instead of the C source's polynomial, every segment against
    EVERY triangle, no tree -- on the Cornell box's depth and normal outputs, every pixel of a 40 x 40 frame at 16 and at 5 samples: the integers are equal (all
    1 600; the test would let a pixel whose answer hangs on 1e-6 of a segment's length or of a triangle's edge lie between numpy's two answers: there is none)"""
    sc = scenes.cornell()
    w = h = 40
    S, cam, recs, out = _f64_frame(orc64, sc, sc.lights, w, h)
    view_inv = np.array(cam.view_inv, np.float64).reshape(4, 4).T
    proj_inv = np.array(cam.proj_inv, np.float64).reshape(4, 4).T
    tris, _, _ = NP.world_triangles(sc.primitives)
    radius = 0.2 * 1.457
    for spp in (16, 5):
        got, st = orc64.render_ao(S, cam, out["depth"], out["normal"], spp, radius, threads=4)
        res = np.array([[NP.ao_pixel(x, y, w, h, float(out["depth"][y, x]), out["normal"][y, x], view_inv, proj_inv, tris, spp, radius) for x in range(w)] for y in range(h)], np.uint32)
        want, lo, hi = res[..., 0], res[..., 1], res[..., 2]
        assert np.all((got >= lo) & (got <= hi)), np.argwhere((got < lo) | (got > hi))[:5]      # every pixel inside what numpy allows it ...
        undecided = lo != hi
        assert np.array_equal(got[~undecided], want[~undecided]) and undecided.mean() < 0.02    # ... which is ONE value for all but a few pixels with a segment ending on a surface
        assert st["ao_rays"] == int((out["depth"] < 10000.0).sum()) * spp
        assert want.min() < 200 and want.max() == 255 and len(np.unique(want)) > spp // 2      # corners and walls: a spread of values, not a constant


def test_presentation_against_numpy_in_fp64(orc):
    """the last leg both hands wrote alike: AMD's LPM tone mapper, restated twice from ffx_lpm.h by the same hand (oracle and art_present.hip).
    tests/np_shading.py restates the presentation a third time, in float64 and from the other side: the control block from the reference's own Rust
    (vk_tonemap.rs:12-47, :122-230, with its z = 1 - x + y), LpmMap from ffx_lpm.h:727-832, the small-float read-back from the format's definition.  4 000 colours
    from black over mid grey to 300 (saturated primaries, near-equal channels, every AO value): the oracle's 8-bit output is numpy's value rounded, or -- where
    numpy's value lies within 0.02 of a rounding boundary and float arithmetic may fall the other way -- its neighbour"""
    rng = np.random.default_rng(7)
    n = 4000
    col = np.zeros((1, n, 4), np.float32)
    mag = 10.0 ** rng.uniform(-4, 2.48, n)
    col[0, :, :3] = (rng.random((n, 3)) ** rng.choice([0.2, 1.0, 4.0], (n, 1))) * mag[:, None]
    col[0, :50, :3] = 0.0                                           # black stays black
    col[0, 50:100, :3] = mag[50:100, None]                          # greys
    col[0, 100:150, 1:3] = 0.0                                      # saturated red
    ao = rng.integers(0, 256, (1, n)).astype(np.uint32); ao[0, :200] = 255
    packed, bgra = orc.present(col, ao)
    P = NP.lpm_setup_709()
    ctl = orc.lpm_control_block(False, 0.0, 256.0, 8.0, 0.25, 1.0, (0, 0, 0), (1.0, 0.5, 1.0 / 32.0)).view(np.float32)
    assert np.allclose(ctl[0:3], P["saturation"], rtol=1e-6) and np.isclose(ctl[3], P["contrast"]) and np.allclose(ctl[4:6], P["tone"], rtol=2e-5)
    assert np.allclose(ctl[6:9], P["luma_t"], rtol=1e-5) and np.allclose(ctl[12:15], P["rcp_luma_t"], rtol=1e-5) and np.allclose(ctl[25:28], P["luma_w"], rtol=1e-5)
    near = exact = 0
    for i in range(n):
        want = NP.present_pixel(int(packed[0, i]), int(ao[0, i]), P)
        got = bgra[0, i, :3].astype(np.float64)
        for k in range(3):
            r = math.floor(want[k] + 0.5)
            if got[k] == r:
                exact += 1
            else:
                assert abs(got[k] - r) == 1 and abs((want[k] + 0.5) - round(want[k] + 0.5)) < 0.02, (i, k, col[0, i], ao[0, i], want, got)
                near += 1
        assert bgra[0, i, 3] == 255
    assert near < 0.01 * 3 * n, (near, exact)
    assert len(np.unique(bgra[0, :, :3])) > 200                      # the whole output range is in the sample
