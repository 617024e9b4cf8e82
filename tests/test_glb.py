"""GLB ingest: the reference-authored known answers of model_reader/gltf_model_reader.rs:684-856 and
model_reader.rs:148-175, restated against libart's reader through the C ABI -- the one part of the pipeline for which the
reference holds real test vectors.  Host only: no GPU needed.  The two .glb fixtures are the reference's own
assets (assets/models/BoxTextured*.glb, data files its tests use), copied to tests/golden/."""
import io
import json
import os
import struct

import numpy as np
import pytest

from araytracingjourney_amd import model_reader as mr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BOX = os.path.join(GOLD, "BoxTextured.glb")
BOX_T = os.path.join(GOLD, "BoxTexturedWithTangents.glb")


# --- gltf_model_reader.rs:690-732
def test_wide_permute_pixel():
    assert mr.permute_pixels([0, 1, 2, 3, 4, 5], 3, {0: 0, 1: 1, 2: 2}, 4).tolist() == [0, 1, 2, 0, 3, 4, 5, 0]


def test_narrow_permute_pixel():
    assert mr.permute_pixels([0, 1, 2, 3, 4, 5, 6, 7], 4, {0: 0, 1: 1, 2: 2}, 3).tolist() == [0, 1, 2, 4, 5, 6]


def test_mix_and_narrow_permute_pixel():
    assert mr.permute_pixels([0, 1, 2, 3, 4, 5, 6, 7], 4, {0: 2, 1: 0, 2: 1}, 3).tolist() == [1, 2, 0, 5, 6, 4]


def test_mix_and_wide_permute_pixel():
    assert mr.permute_pixels([0, 1, 2, 3, 4, 5], 3, {0: 2, 1: 0, 2: 1}, 4).tolist() == [1, 2, 0, 0, 4, 5, 3, 0]


def test_mix_same_size_permute_pixel():
    """:734-750 compares the scalar permute with its SSSE3/AVX2 twins; there is one implementation here, checked against numpy"""
    src = np.arange(128, dtype=np.uint8)
    got = mr.permute_pixels(src, 4, {0: 2, 1: 0, 2: 1, 3: 3}, 4)
    want = src.reshape(-1, 4)[:, [1, 2, 0, 3]].reshape(-1)
    assert np.array_equal(got, want)


# --- gltf_model_reader.rs:784-855 test_textured_cube
def test_textured_cube():
    r = mr.GltfModelReader(BOX, True, mr.COERCE_B8G8R8A8)
    centre, radius = r.get_primitives_bounding_sphere()
    assert (radius - 1.0) < 1e-5 and np.all((centre - 1.0) < 1e-5)                 # the reference's (one-sided) assertions
    assert abs(radius - math_sqrt(0.75)) < 1e-6 and np.allclose(centre, 0, atol=1e-6)  # what Ritter's sphere of a +-0.5 cube is
    data, infos = r.copy_model_data_to_ptr(mr.VERTICES | mr.NORMALS | mr.TEX_COORDS | mr.INDICES, mr.ALBEDO)
    ci = infos[0]
    assert data.size == ci.mesh_size + ci.indices_size + ci.image_size                # compute_total_size (model_reader.rs:85-93)
    first_vertex = data[ci.mesh_buffer_offset:ci.mesh_buffer_offset + 32].view(np.float32)
    assert np.allclose(first_vertex, [-0.5, -0.5, 0.5, 6.0, 0.0, 0.0, 0.0, 1.0], atol=1e-7)   # pos, uv, normal: the bitflag order
    assert data[ci.indices_buffer_offset:ci.indices_buffer_offset + 8].view(np.uint16).tolist() == [0, 1, 2, 3]
    assert data[ci.image_buffer_offset:ci.image_buffer_offset + 4].tolist() == [220, 220, 220, 0]  # RGB -> BGRA widened, alpha 0
    assert (ci.single_mesh_element_size, ci.single_index_size, ci.image_width, ci.image_height, ci.image_layers, ci.image_mip_levels) == (32, 2, 256, 256, 1, 1)
    assert ci.image_format == 5 and ci.mesh_size == 24 * 32 and ci.indices_size == 72   # B8G8R8A8; 24 vertices, 36 indices


def math_sqrt(x):
    return float(np.sqrt(np.float64(x)))


def test_interleave_order_is_the_bitflag_order():
    """model_reader.rs:148-175: VERTICES, TEX_COORDS, NORMALS, TANGENTS, (INDICES) -- the 48-byte vertex of the ray tracer"""
    r = mr.GltfModelReader(BOX_T, True, mr.COERCE_B8G8R8A8)
    data, infos = r.copy_model_data_to_ptr(mr.VERTICES | mr.TEX_COORDS | mr.NORMALS | mr.TANGENTS | mr.INDICES, mr.ALBEDO)
    ci = infos[0]
    assert ci.single_mesh_element_size == 48 and ci.mesh_size == 24 * 48
    v = data[ci.mesh_buffer_offset:ci.mesh_buffer_offset + ci.mesh_size].view(np.float32).reshape(24, 12)
    doc, bin_ = _parse(BOX_T)
    acc = {k: _accessor(doc, bin_, i) for k, i in doc["meshes"][0]["primitives"][0]["attributes"].items()}
    assert np.array_equal(v[:, 0:3], acc["POSITION"]) and np.array_equal(v[:, 3:5], acc["TEXCOORD_0"])
    assert np.array_equal(v[:, 5:8], acc["NORMAL"]) and np.array_equal(v[:, 8:12], acc["TANGENT"])
    only_nrm_idx, _ = r.copy_model_data_to_ptr(mr.NORMALS | mr.INDICES, 0)
    assert only_nrm_idx.size == 24 * 12 + 72


def test_png_decode_matches_pillow():
    """independent pin of the PNG path (palette image): Pillow's decode of the same embedded stream"""
    from PIL import Image
    doc, bin_ = _parse(BOX)
    bv = doc["bufferViews"][doc["images"][0]["bufferView"]]
    ref = np.asarray(Image.open(io.BytesIO(bin_[bv["byteOffset"]:bv["byteOffset"] + bv["byteLength"]])).convert("RGB"))
    r = mr.GltfModelReader(BOX, False, mr.COERCE_NONE)
    data, infos = r.copy_model_data_to_ptr(0, mr.ALBEDO)
    assert infos[0].image_format == 2 and infos[0].image_size == 256 * 256 * 3          # R8G8B8: palettes expand to RGB
    assert np.array_equal(data[infos[0].image_buffer_offset:].reshape(256, 256, 3), ref)
    rgba = mr.GltfModelReader(BOX, False, mr.COERCE_R8G8B8A8).copy_model_data_to_ptr(0, mr.ALBEDO)[0].reshape(256, 256, 4)
    assert np.array_equal(rgba[..., :3], ref) and (rgba[..., 3] == 0).all()
    bgra = mr.GltfModelReader(BOX, False, mr.COERCE_B8G8R8A8).copy_model_data_to_ptr(0, mr.ALBEDO)[0].reshape(256, 256, 4)
    assert np.array_equal(bgra[..., [2, 1, 0]], ref)


def test_jpeg_textures_decode_close_to_pillow(tmp_path, get_scene):
    """JPEG in a GLB (the gltf crate's import() decodes JPEG as well as PNG): baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0, grey, restart intervals.
    Decoders differ by an LSB or two in the inverse DCT and the chroma filter: mean |diff| < 0.6, max <= 6 against Pillow (libjpeg-turbo)"""
    from PIL import Image
    from glb_writer import write_glb
    from araytracingjourney_amd import scenes
    from araytracingjourney_amd._lib import ArtError
    sc = get_scene("sponza_like", 0.05)
    p = sc.primitives[0]
    yy, xx = np.mgrid[0:120, 0:200].astype(np.float32)            # a smooth picture with some structure, sizes that are no multiple of 16
    pic = np.stack([127 + 120 * np.sin(xx / 17) * np.cos(yy / 23), 127 + 100 * np.cos(xx / 9 + yy / 31), 40 + xx * 0.9 + 20 * np.sin(yy / 5), 255 + 0 * xx], -1)
    tex = np.broadcast_to(np.clip(pic, 0, 255).astype(np.uint8), (3, 120, 200, 4)).copy()
    prim = scenes.Primitive(p.verts, p.indices, tex, p.model)
    for modes in (("JPEG444", "JPEG420", "JPEG422"), ("JPEG420+R3", "JPEG444+R1", "JPEG422+R7"), ("JPEG444P", "JPEG420P", "JPEG422P"), ("JPEG420P+R2", "JPEG444P+R5", "JPEG422P")):
        path = tmp_path / ("j_" + "_".join(m.replace("+", "") for m in modes) + ".glb")
        write_glb(str(path), [prim], png_modes=modes)
        doc, bin_ = _parse(str(path))
        r = mr.GltfModelReader(str(path), True, mr.COERCE_R8G8B8A8)
        data, infos = r.copy_model_data_to_ptr(0, mr.ALBEDO | mr.ORM | mr.NORMAL)
        got = data[infos[0].image_buffer_offset:infos[0].image_buffer_offset + infos[0].image_size].reshape(3, 120, 200, 4)
        for layer in range(3):
            bv = doc["bufferViews"][doc["images"][layer]["bufferView"]]
            ref = np.asarray(Image.open(io.BytesIO(bin_[bv["byteOffset"]:bv["byteOffset"] + bv["byteLength"]])).convert("RGB")).astype(np.int32)
            d = np.abs(got[layer][..., :3].astype(np.int32) - ref)
            assert d.mean() < 0.6 and d.max() <= 6, (modes[layer], float(d.mean()), int(d.max()))
    # a grey JPEG decodes to R8: readable without coercion, and the coercion panics on it exactly like on a grey PNG (gltf_model_reader.rs:485)
    path = tmp_path / "grey.glb"
    write_glb(str(path), [prim], png_modes=("JPEGL", "JPEGL", "JPEGL"))
    doc, bin_ = _parse(str(path))
    data, infos = mr.GltfModelReader(str(path), True, mr.COERCE_NONE).copy_model_data_to_ptr(0, mr.ALBEDO)
    bv = doc["bufferViews"][doc["images"][0]["bufferView"]]
    ref = np.asarray(Image.open(io.BytesIO(bin_[bv["byteOffset"]:bv["byteOffset"] + bv["byteLength"]]))).astype(np.int32)
    got = data[infos[0].image_buffer_offset:infos[0].image_buffer_offset + infos[0].image_size].reshape(120, 200).astype(np.int32)
    assert infos[0].image_format == 0 and np.abs(got - ref).max() <= 2
    with pytest.raises(ArtError, match="Unsupported source format"):
        mr.GltfModelReader(str(path), True, mr.COERCE_R8G8B8A8).copy_model_data_to_ptr(0, mr.ALBEDO)



def test_missing_attributes_and_textures_are_errors_like_the_reference_panics():
    from araytracingjourney_amd._lib import ArtError
    r = mr.GltfModelReader(BOX, True, mr.COERCE_B8G8R8A8)
    with pytest.raises(ArtError, match="not found"):
        r.copy_model_data_to_ptr(mr.VERTICES | mr.TANGENTS, 0)        # BoxTextured has no tangents
    with pytest.raises(ArtError, match="not found"):
        r.copy_model_data_to_ptr(mr.VERTICES, mr.ALBEDO | mr.ORM)      # nor an ORM texture (gltf_model_reader.rs:261-263)
    with pytest.raises(ArtError, match="Could not read file"):
        mr.GltfModelReader("/nonexistent.glb")


def test_synthetic_glb_round_trip(tmp_path, get_scene):
    """normalisation (:415-460), strided views, RGB/palette/RGBA PNGs, u16 + u32 indices, several primitives"""
    from glb_writer import write_glb
    from araytracingjourney_amd import scenes
    sc = get_scene("sponza_like", 0.05)
    prims = [sc.primitives[0], sc.primitives[6], sc.primitives[24]]
    big = [scenes.Primitive((p.verts * np.array([3.0, 3.0, 3.0] + [1.0] * 9, np.float32)).astype(np.float32), p.indices, p.tex[:, ::8, ::8].copy(), p.model) for p in prims]
    for interleaved in (False, True):
        path = tmp_path / f"m{int(interleaved)}.glb"
        write_glb(str(path), big, png_modes=("RGBA", "RGB", "RGBA"), interleaved=interleaved)
        r = mr.GltfModelReader(str(path), True, mr.COERCE_B8G8R8A8)
        assert r.primitive_count() == 3
        data, infos = r.copy_model_data_to_ptr(mr.VERTICES | mr.TEX_COORDS | mr.NORMALS | mr.TANGENTS | mr.INDICES, mr.ALBEDO | mr.ORM | mr.NORMAL)
        mx = max(np.sqrt((p.verts[:, :3].astype(np.float32) ** 2).sum(1, dtype=np.float32)).max() for p in big)
        assert mx > 1
        for p, ci in zip(big, infos):
            v = data[ci.mesh_buffer_offset:ci.mesh_buffer_offset + ci.mesh_size].view(np.float32).reshape(-1, 12)
            assert np.array_equal(v[:, 3:], p.verts[:, 3:])
            assert np.array_equal(v[:, :3], p.verts[:, :3] / np.float32(mx))                      # every primitive by the global max magnitude
            assert ci.single_index_size == p.indices.dtype.itemsize
            assert np.array_equal(data[ci.indices_buffer_offset:ci.indices_buffer_offset + ci.indices_size].view(p.indices.dtype), p.indices)
            tex = data[ci.image_buffer_offset:ci.image_buffer_offset + ci.image_size].reshape(3, p.tex.shape[1], p.tex.shape[2], 4)
            assert ci.image_layers == 3 and ci.image_buffer_offset % 4 == 0
            assert np.array_equal(tex[0][..., [2, 1, 0, 3]], p.tex[0]) and np.array_equal(tex[2][..., [2, 1, 0, 3]], p.tex[2])
            assert np.array_equal(tex[1][..., [2, 1, 0]], p.tex[1][..., :3]) and (tex[1][..., 3] == 0).all()  # RGB source: alpha 0 after widening
        c, rad = r.get_primitives_bounding_sphere()
        allv = np.concatenate([p.verts[:, :3] / np.float32(mx) for p in big])
        assert (np.linalg.norm(allv - c, axis=1) <= rad * (1 + 1e-5)).all() and rad <= 1.2
    path = tmp_path / "two_meshes.glb"
    write_glb(str(path), big[:1], n_meshes=2)
    from araytracingjourney_amd._lib import ArtError
    with pytest.raises(ArtError, match="exactly one mesh"):
        mr.GltfModelReader(str(path))


def _parse(path):
    b = open(path, "rb").read()
    clen, _ = struct.unpack("<II", b[12:20])
    doc = json.loads(b[20:20 + clen])
    off = 20 + clen
    blen, _ = struct.unpack("<II", b[off:off + 8])
    return doc, b[off + 8:off + 8 + blen]


def _accessor(doc, bin_, i):
    a = doc["accessors"][i]
    v = doc["bufferViews"][a["bufferView"]]
    n = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}[a["type"]]
    off = v.get("byteOffset", 0) + a.get("byteOffset", 0)
    stride = v.get("byteStride", n * 4)
    out = np.zeros((a["count"], n), np.float32)
    for k in range(a["count"]):
        out[k] = np.frombuffer(bin_, np.float32, n, off + k * stride)
    return out


# ---- malformed files: the reference indexes Rust slices and panics; a C++ reader must answer with an error code, never touch
# memory outside the file.  Every case runs twice: through libart.so (the product build) and through a g++ AddressSanitizer +
# UBSan build of the same sources (tests/glb_asan_driver.cpp), where an out-of-bounds byte is a report and a non-zero exit.
def _glb_bytes(doc, bin_):
    js = (json.dumps(doc) if not isinstance(doc, (bytes, str)) else doc)
    js = js.encode() if isinstance(js, str) else js
    js += b" " * (-len(js) % 4)
    bin_ = bytes(bin_) + b"\0" * (-len(bin_) % 4)
    return struct.pack("<III", 0x46546C67, 2, 28 + len(js) + len(bin_)) + struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(bin_), 0x004E4942) + bin_


def _tri_doc(**over):
    """one triangle: POSITION vec3 f32 x 3 at 0, indices u16 x 3 at 36; `over` patches accessor / view fields"""
    doc = {"asset": {"version": "2.0"}, "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]}], "buffers": [{"byteLength": 44}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 6}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}]}
    for path, v in over.items():
        tab, i, key = path.split("__")
        if v is None:
            doc[tab][int(i)].pop(key, None)
        else:
            doc[tab][int(i)][key] = v
    return doc


_TRI_BIN = np.array([0, 0, 0, 1, 0, 0, 0, 1, 0], np.float32).tobytes() + np.array([0, 1, 2], np.uint16).tobytes()


def _png_header_only(w, h):
    import zlib
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")


def _jpeg_header_only(w, h):
    sof = struct.pack(">BBHBHHB", 0xFF, 0xC0, 17, 8, h, w, 3) + bytes([1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1])
    return b"\xFF\xD8" + sof + b"\xFF\xDA" + struct.pack(">H", 12) + bytes([3, 1, 0, 2, 0x11, 3, 0x11, 0, 63, 0]) + b"\0" * 16 + b"\xFF\xD9"


def _image_doc(payload):
    doc = _tri_doc()
    doc["bufferViews"].append({"buffer": 0, "byteOffset": 44, "byteLength": len(payload)})
    doc["images"] = [{"bufferView": 2, "mimeType": "image/png"}]
    doc["textures"] = [{"source": 0}]
    doc["materials"] = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}]
    doc["meshes"][0]["primitives"][0]["material"] = 0
    return doc, _TRI_BIN + b"\0\0" + payload


MALFORMED = {
    # byteStride 4 < the 12-byte element: (stride - elem_size) used to wrap to 0xFFFFFFF8 and wave any offset through
    "stride_smaller_than_element": (_tri_doc(bufferViews__0__byteStride=4, accessors__0__byteOffset=1 << 20), _TRI_BIN, "byteStride"),
    # a SCALAR/u8 POSITION accessor: normalize_vectors used to copy 12 bytes per element before validate_model had looked
    "scalar_u8_position": (_tri_doc(accessors__0__type="SCALAR", accessors__0__componentType=5121, accessors__0__count=44), _TRI_BIN, "element size"),
    "window_past_the_end": (_tri_doc(accessors__0__count=4), _TRI_BIN, "out of range"),
    "offset_past_the_end": (_tri_doc(accessors__0__byteOffset=40), _TRI_BIN, "out of range"),
    "huge_offset_wraps": (_tri_doc(accessors__0__byteOffset=2 ** 53, bufferViews__0__byteOffset=2 ** 53), _TRI_BIN, "out of range"),
    "offset_beyond_double_integers": (_tri_doc(accessors__0__byteOffset=1e300), _TRI_BIN, "non-negative integers"),
    "negative_offset": (_tri_doc(bufferViews__0__byteOffset=-8), _TRI_BIN, "non-negative integers"),
    "negative_count": (_tri_doc(accessors__0__count=-3), _TRI_BIN, "non-negative integers"),
    "fractional_count": (_tri_doc(accessors__0__count=2.5), _TRI_BIN, "non-negative integers"),
    "zero_count": (_tri_doc(accessors__0__count=0), _TRI_BIN, "count is zero"),
    "count_times_stride_wraps": (_tri_doc(accessors__0__count=2 ** 52, bufferViews__0__byteStride=65536), _TRI_BIN, "out of range"),
    "index_window_past_the_end": (_tri_doc(accessors__1__count=5), _TRI_BIN, "out of range"),
    "accessor_index_negative": ({**_tri_doc(), "meshes": [{"primitives": [{"attributes": {"POSITION": -1}, "indices": 1}]}]}, _TRI_BIN, "accessor index"),
    "accessor_index_huge": ({**_tri_doc(), "meshes": [{"primitives": [{"attributes": {"POSITION": 7}, "indices": 1}]}]}, _TRI_BIN, "accessor index"),
    "buffer_view_index_out_of_range": (_tri_doc(accessors__0__bufferView=9), _TRI_BIN, "buffer view"),
    "no_component_type": (_tri_doc(accessors__0__componentType=None), _TRI_BIN, "accessor type"),
    "deeply_nested_json": ('{"asset":' + "[" * 100000 + "]" * 100000 + "}", _TRI_BIN, "does not parse"),
    "unterminated_unicode_escape": ('{"asset":"\\u12', _TRI_BIN, "does not parse"),
    "image_view_out_of_range": (lambda: (lambda d, b: (dict(d, bufferViews=d["bufferViews"][:2] + [{"buffer": 0, "byteOffset": 40, "byteLength": 1 << 30}]), b))(*_image_doc(b"x")), None, "image buffer view"),
    "image_view_negative_length": (lambda: (lambda d, b: (dict(d, bufferViews=d["bufferViews"][:2] + [{"buffer": 0, "byteOffset": 44, "byteLength": -1}]), b))(*_image_doc(b"x")), None, "non-negative integers"),
    "png_of_4_gigapixels": (lambda: _image_doc(_png_header_only(65535, 65535)), None, "extent"),
    "png_truncated_pixel_data": (lambda: _image_doc(_png_header_only(64, 64)), None, "inflate"),
    "jpeg_of_4_gigapixels": (lambda: _image_doc(_jpeg_header_only(65535, 65535)), None, "extent"),
    "jpeg_without_tables": (lambda: _image_doc(_jpeg_header_only(16, 16)), None, "table"),
    "truncated_chunk": (None, None, "truncated"),
}


def _malformed_file(tmp_path, name):
    doc, bin_, _ = MALFORMED[name]
    if name == "truncated_chunk":
        data = _glb_bytes(_tri_doc(), _TRI_BIN)[:-20]
    else:
        if callable(doc):
            doc, bin_ = doc()
        data = _glb_bytes(doc, bin_)
    path = tmp_path / (name + ".glb")
    path.write_bytes(data)
    return str(path)


@pytest.fixture(scope="module")
def asan_driver(tmp_path_factory):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path_factory.mktemp("asan") / "glb_asan")
    src = [os.path.join(root, "araytracingjourney_amd", "csrc", f) for f in ("art_glb.hip", "art_jpeg.hip")] + [os.path.join(root, "tests", "glb_asan_driver.cpp")]
    subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] + src + ["-lz", "-o", exe])
    return exe


def test_the_well_formed_triangle_reads(tmp_path):
    """the base document the malformed cases are patches of"""
    path = tmp_path / "tri.glb"
    path.write_bytes(_glb_bytes(_tri_doc(), _TRI_BIN))
    r = mr.GltfModelReader(str(path), True, mr.COERCE_NONE)
    data, infos = r.copy_model_data_to_ptr(mr.VERTICES | mr.INDICES, 0)
    assert data[:36].view(np.float32).tolist() == [0, 0, 0, 1, 0, 0, 0, 1, 0] and data[36:42].view(np.uint16).tolist() == [0, 1, 2]


@pytest.mark.parametrize("name", sorted(MALFORMED))
def test_malformed_glb_is_an_error_not_a_memory_fault(tmp_path, asan_driver, name):
    import subprocess
    from araytracingjourney_amd._lib import ArtError
    path = _malformed_file(tmp_path, name)
    with pytest.raises(ArtError, match=MALFORMED[name][2]):
        mr.GltfModelReader(path, True, mr.COERCE_B8G8R8A8)
    out = subprocess.run([asan_driver, path], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "rc=-" in out.stdout and "ERROR" not in out.stderr, out.stdout + out.stderr


def test_wellformed_assets_are_clean_under_the_sanitizers(tmp_path, asan_driver, get_scene):
    """the reference's two boxes and a synthetic strided / JPEG / palette file through the sanitizer build: every path, no report"""
    import subprocess
    from glb_writer import write_glb
    sc = get_scene("sponza_like", 0.05)
    p = sc.primitives[0]
    from araytracingjourney_amd import scenes
    small = scenes.Primitive(p.verts, p.indices, p.tex[:, ::8, ::8].copy(), p.model)
    files = [BOX, BOX_T]
    for i, (modes, inter) in enumerate(((("RGBA", "RGB", "P"), True), (("JPEG420", "JPEG444P", "JPEG422+R3"), False))):
        path = tmp_path / f"ok{i}.glb"
        write_glb(str(path), [small], png_modes=modes, interleaved=inter)
        files.append(str(path))
    for f in files:
        out = subprocess.run([asan_driver, f], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and out.stdout.startswith("rc=0") and "ERROR" not in out.stderr, (f, out.stdout + out.stderr)


def test_image_offset_alignment_is_integer_arithmetic():
    """align_offset (model_reader.rs:144-146) in f32 moves an offset above 2^24 bytes BACKWARDS into the index data; ours is integer.
    A primitive with 1.4 M vertices puts its texture past 64 MB (sizing pass only: nothing is copied)."""
    n = 1_400_001                                        # 1 400 001 * 48 = 67 200 048 bytes of vertices: past 2^26
    doc = {"asset": {"version": "2.0"}, "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1, "NORMAL": 2, "TANGENT": 3}, "indices": 4, "material": 0}]}],
           "buffers": [{"byteLength": 0}], "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": n * 48, "byteStride": 48}, {"buffer": 0, "byteOffset": n * 48, "byteLength": 18}],
           "accessors": [{"bufferView": 0, "byteOffset": o, "componentType": 5126, "count": n, "type": t} for o, t in ((0, "VEC3"), (12, "VEC2"), (20, "VEC3"), (32, "VEC4"))]
                        + [{"bufferView": 1, "componentType": 5123, "count": 9, "type": "SCALAR"}]}
    import tempfile
    from PIL import Image
    buf = io.BytesIO(); Image.fromarray(np.full((2, 2, 4), 200, np.uint8), "RGBA").save(buf, format="PNG"); png = buf.getvalue()
    bin_ = bytearray(n * 48) + bytearray(np.array([0, 1, 2] * 3, np.uint16).tobytes()) + b"\0\0" + png
    doc["bufferViews"].append({"buffer": 0, "byteOffset": n * 48 + 20, "byteLength": len(png)})
    doc["images"] = [{"bufferView": 2}]; doc["textures"] = [{"source": 0}]
    doc["materials"] = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicRoughnessTexture": {"index": 0}}, "normalTexture": {"index": 0}}]
    with tempfile.NamedTemporaryFile(suffix=".glb") as f:
        f.write(_glb_bytes(doc, bin_)); f.flush()
        r = mr.GltfModelReader(f.name, False, mr.COERCE_R8G8B8A8)
        _, infos = r.copy_model_data_to_ptr(mr.VERTICES | mr.TEX_COORDS | mr.NORMALS | mr.TANGENTS | mr.INDICES, mr.ALBEDO | mr.ORM | mr.NORMAL, copy=False)
    ci = infos[0]
    end_of_indices = ci.indices_buffer_offset + ci.indices_size
    assert end_of_indices == n * 48 + 18 and end_of_indices > 1 << 26
    assert ci.image_buffer_offset == (end_of_indices + 3) // 4 * 4                       # the next multiple of the 4-byte texel
    f32 = int(np.float32(4) * np.ceil(np.float32(end_of_indices) / np.float32(4)))       # what the reference's f32 expression gives
    assert f32 < end_of_indices                                                          # ... which is inside the index data: the latent bug
