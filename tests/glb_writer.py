"""Test helper: writes a GLB 2.0 file (one mesh, one buffer, PNG images) from scenes.Primitive objects, so that the GLB
ingest can be exercised on files with tangents, ORM and normal textures (the reference ships only two albedo-only boxes)."""
import io
import json
import struct

import numpy as np
from PIL import Image


def _png(arr, mode=None):
    buf = io.BytesIO()
    img = Image.fromarray(arr, mode) if mode else Image.fromarray(arr)
    img.save(buf, format="PNG")
    return buf.getvalue()


def write_glb(path, primitives, png_modes=("RGBA", "RGB", "RGBA"), interleaved=False, n_meshes=1):
    bin_ = bytearray()
    views, accessors, images, textures, materials, prims = [], [], [], [], [], []

    def add_view(data, stride=None):
        while len(bin_) % 4:
            bin_.append(0)
        v = {"buffer": 0, "byteOffset": len(bin_), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        bin_.extend(data)
        views.append(v)
        return len(views) - 1

    for p in primitives:
        v = p.verts.astype(np.float32)
        n = v.shape[0]
        attrs = {}
        if interleaved:   # one strided view for all four attributes
            vi = add_view(np.ascontiguousarray(v).tobytes(), 48)
            for name, off, typ in (("POSITION", 0, "VEC3"), ("TEXCOORD_0", 12, "VEC2"), ("NORMAL", 20, "VEC3"), ("TANGENT", 32, "VEC4")):
                accessors.append({"bufferView": vi, "byteOffset": off, "componentType": 5126, "count": n, "type": typ})
                attrs[name] = len(accessors) - 1
        else:
            for name, sl, typ in (("POSITION", slice(0, 3), "VEC3"), ("TEXCOORD_0", slice(3, 5), "VEC2"), ("NORMAL", slice(5, 8), "VEC3"), ("TANGENT", slice(8, 12), "VEC4")):
                vi = add_view(np.ascontiguousarray(v[:, sl]).tobytes())
                accessors.append({"bufferView": vi, "componentType": 5126, "count": n, "type": typ})
                attrs[name] = len(accessors) - 1
        idx = p.indices
        vi = add_view(idx.tobytes())
        accessors.append({"bufferView": vi, "componentType": 5123 if idx.dtype == np.uint16 else 5125, "count": int(idx.size), "type": "SCALAR"})
        ia = len(accessors) - 1
        tex_ids = []
        for layer, mode in zip(range(3), png_modes):
            px = p.tex[layer]
            if mode == "RGB":
                data = _png(np.ascontiguousarray(px[..., :3]), "RGB")
            elif mode == "P":
                data = io.BytesIO()
                Image.fromarray(np.ascontiguousarray(px[..., :3]), "RGB").quantize(256).save(data, format="PNG")
                data = data.getvalue()
            elif mode.startswith("JPEG"):   # "JPEG444", "JPEG420", "JPEG422", "JPEGL" (grey), optional "+R<n>" restart interval in MCUs
                data = io.BytesIO()
                kind, _, rst = mode.partition("+R")
                img = Image.fromarray(np.ascontiguousarray(px[..., :3]), "RGB")
                kw = dict(format="JPEG", quality=92, progressive=kind.endswith("P"))
                kind = kind[:-1] if kind.endswith("P") else kind
                if kind == "JPEGL":
                    img = img.convert("L")
                else:
                    kw["subsampling"] = {"JPEG444": 0, "JPEG422": 1, "JPEG420": 2}[kind]
                if rst:
                    kw["restart_marker_blocks"] = int(rst)
                img.save(data, **kw)
                data = data.getvalue()
            else:
                data = _png(np.ascontiguousarray(px), "RGBA")
            vi = add_view(data)
            images.append({"bufferView": vi, "mimeType": "image/jpeg" if mode.startswith("JPEG") else "image/png"})
            textures.append({"source": len(images) - 1})
            tex_ids.append(len(textures) - 1)
        materials.append({"pbrMetallicRoughness": {"baseColorTexture": {"index": tex_ids[0]}, "metallicRoughnessTexture": {"index": tex_ids[1]}},
                          "normalTexture": {"index": tex_ids[2]}})
        prims.append({"attributes": attrs, "indices": ia, "material": len(materials) - 1})
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
           "meshes": [{"primitives": prims}] * n_meshes, "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": len(bin_)}],
           "images": images, "textures": textures, "materials": materials}
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    while len(bin_) % 4:
        bin_.append(0)
    total = 12 + 8 + len(js) + 8 + len(bin_)
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, total))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(bin_), 0x004E4942) + bytes(bin_))
