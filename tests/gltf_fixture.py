"""Writes a small GLB that exercises the loader's cases (no assets ship with the reference: SURVEY F4)."""
import io
import json
import struct

import numpy as np


def _png(arr, mode):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(arr, mode).save(b, format="PNG")
    return b.getvalue()


def write_test_glb(path):
    """Returns what the loader must find.  Layout:
    node 0 (root, matrix: translate (1,2,3))
      node 1 (TRS: rotate 90 deg about Y, scale 2, translate (0,0,-1)) -> mesh 0: 2 primitives
         prim 0: quad, u16 indices, normals, TEXCOORD_0 as normalised u8, material 1
         prim 1: one triangle, no indices/normals/uvs, material 0, interleaved (byteStride) positions
         (+ a LINES primitive that must be skipped)
    node 2 (camera, matrix)   node 3 (mesh 0 again, outside the scene: identity)
    images: 0 = RGB linear (normal map), 1 = RGBA sRGB (base colour), 2 = greyscale (-> 0.5)
    textures: 0 -> image 1, 1 -> image 0, 2 -> image 2     materials: 0 plain, 1 textured + transmission + ior
    """
    rng = np.random.default_rng(5)
    blob = bytearray()
    views, accessors = [], []

    def add_view(data, stride=None):
        while len(blob) % 4:
            blob.append(0)
        v = {"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        blob.extend(data)
        views.append(v)
        return len(views) - 1

    def add_acc(view, ctype, count, typ, offset=0, normalized=False):
        a = {"bufferView": view, "componentType": ctype, "count": count, "type": typ, "byteOffset": offset}
        if normalized:
            a["normalized"] = True
        accessors.append(a)
        return len(accessors) - 1

    quad_p = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], dtype=np.float32)
    quad_n = np.tile(np.array([[0, 1, 0]], dtype=np.float32), (4, 1))
    quad_uv = np.array([[0, 0], [255, 0], [255, 255], [0, 255]], dtype=np.uint8)
    quad_i = np.array([0, 2, 1, 0, 3, 2], dtype=np.uint16)
    a_p = add_acc(add_view(quad_p.tobytes()), 5126, 4, "VEC3")
    a_n = add_acc(add_view(quad_n.tobytes()), 5126, 4, "VEC3")
    a_uv = add_acc(add_view(quad_uv.tobytes()), 5121, 4, "VEC2", normalized=True)
    a_i = add_acc(add_view(quad_i.tobytes()), 5123, 6, "SCALAR")
    tri_p = np.array([[0, 0.5, 0], [0.5, 1.5, 0], [-0.5, 1.5, 0.25]], dtype=np.float32)
    inter = np.zeros((3, 6), dtype=np.float32)          # position + 12 bytes of padding per vertex: byteStride 24
    inter[:, :3] = tri_p
    inter[:, 3:] = 99.0
    a_tp = add_acc(add_view(inter.tobytes(), stride=24), 5126, 3, "VEC3")

    img_lin = rng.integers(0, 256, size=(4, 8, 3), dtype=np.uint8)
    img_srgb = rng.integers(0, 256, size=(8, 4, 4), dtype=np.uint8)
    img_grey = rng.integers(0, 256, size=(2, 2), dtype=np.uint8)
    v_img = [add_view(_png(img_lin, "RGB")), add_view(_png(img_srgb, "RGBA")), add_view(_png(img_grey, "L"))]

    root_m = np.eye(4, dtype=np.float32); root_m[:3, 3] = [1, 2, 3]
    cam_m = np.eye(4, dtype=np.float32); cam_m[:3, 3] = [0, 1.5, 6]
    s = float(np.sin(np.pi / 4))
    doc = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 2]}],
        "nodes": [
            {"matrix": root_m.T.reshape(-1).tolist(), "children": [1]},
            {"mesh": 0, "rotation": [0.0, s, 0.0, s], "scale": [2.0, 2.0, 2.0], "translation": [0.0, 0.0, -1.0]},
            {"camera": 0, "matrix": cam_m.T.reshape(-1).tolist()},
            {"mesh": 0},
        ],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.8, "znear": 0.1}}],
        "meshes": [{"primitives": [
            {"attributes": {"POSITION": a_p, "NORMAL": a_n, "TEXCOORD_0": a_uv}, "indices": a_i, "material": 1},
            {"attributes": {"POSITION": a_tp}},
            {"attributes": {"POSITION": a_p}, "mode": 1},
        ]}],
        "materials": [
            {"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.7, 0.6, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.5}},
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicRoughnessTexture": {"index": 2}},
             "normalTexture": {"index": 1, "scale": 0.75}, "emissiveFactor": [0.1, 0.2, 0.3],
             "extensions": {"KHR_materials_transmission": {"transmissionFactor": 0.25}, "KHR_materials_ior": {"ior": 1.33}}},
        ],
        "textures": [{"source": 1}, {"source": 0}, {"source": 2}],
        "images": [{"bufferView": v, "mimeType": "image/png"} for v in v_img],
        "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": len(blob)}],
    }
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    while len(blob) % 4:
        blob.append(0)
    total = 12 + 8 + len(js) + 8 + len(blob)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, total))
        f.write(struct.pack("<II", len(js), 0x4E4F534A)); f.write(js)
        f.write(struct.pack("<II", len(blob), 0x004E4942)); f.write(bytes(blob))
    return {"quad_p": quad_p, "quad_uv": quad_uv, "quad_i": quad_i, "tri_p": tri_p, "img_lin": img_lin, "img_srgb": img_srgb,
            "root_m": root_m, "cam_m": cam_m}
