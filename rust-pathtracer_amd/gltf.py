"""glTF 2.0 / GLB -> flat scene: what `load_gltf` + `Scene::new` hand to the render core.

Mirrors `src/raytracer/loader/gltf.rs`:
  * one Blas + one instance per TRIANGLES primitive, in document node order, with the node's global transform
    (`load_meshes` :237-309, `compute_global_transforms` :210-235: parent * local, scenes' roots walked in order;
    nodes outside every scene keep the identity);
  * missing indices => 0,1,2,...; missing normals / uvs stay empty; TEXCOORD_0 through `into_f32` (u8/255, u16/65535);
  * one texture per IMAGE (not per glTF texture); materials store image indices through the texture->image map
    (:149-156); images referenced as base colour or emissive are converted sRGB -> linear (:108-133, :42-48);
    R8G8B8 / R8G8B8A8 / 16-bit RGB(A) decode as in :50-106, every other pixel format becomes flat 0.5 grey (:103);
  * materials: pbrMetallicRoughness factors and textures, normal scale, emissive, KHR_materials_transmission,
    KHR_materials_ior (:158-207); no materials => one `Material::default()` (main.rs:104-107);
  * the first perspective camera in node order: position = column 3, forward = -normalize(column 2),
    fov = yfov in degrees (:311-338).

Image files are decoded with PIL, the reference with the `image` crate: PARITY UNPINNED for the texel values.  PNG
decoding is lossless, so 8-bit RGB(A) PNGs give the same pixels; JPEG does not define its IDCT bit for bit, so PIL's
(libjpeg) texels can differ from the `image` crate's by a few levels; PIL hands 16-bit RGB PNGs over as 8-bit.  The
loader says so per image in `GltfScene.notes` (JPEG and 16-bit cases, like the grey formats).  Node TRS -> matrix uses the usual
T * R(q) * S in float32; the `gltf` crate's exact operation order is not available here, so transforms of TRS nodes are
"parity unpinned" to the last ulp — nodes that carry a `matrix` pass through bit for bit.
"""
import base64
import ctypes
import io
import json
import os
import struct
from dataclasses import dataclass, field

import numpy as np

from . import host

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class GltfError(ValueError):
    pass


@dataclass
class GltfMesh:
    positions: np.ndarray            # (n, 3) f32
    normals: np.ndarray              # (n, 3) f32 or (0, 3)
    uvs: np.ndarray                  # (n, 2) f32 or (0, 2)
    indices: np.ndarray              # (t, 3) u32
    material_id: int


@dataclass
class GltfScene:
    meshes: list = field(default_factory=list)        # one per primitive == `blases`
    instances: list = field(default_factory=list)     # (blas_index, 4x4 f32, row-major numpy == column vectors M @ p)
    materials: list = field(default_factory=list)     # PtxMaterial
    textures: list = field(default_factory=list)      # (h, w, 3) f32, one per glTF image
    camera: dict = None                               # {"position", "look_at", "fov_degrees"} or None
    notes: list = field(default_factory=list)


def _srgb_lut():
    """srgb_to_linear (gltf.rs:42-48) for c = i/255, evaluated with glibc powf like rustc's f32::powf."""
    c = np.arange(256, dtype=np.float32) / np.float32(255.0)
    out = (c / np.float32(12.92)).astype(np.float32)
    try:
        libm = ctypes.CDLL("libm.so.6")
        libm.powf.restype = ctypes.c_float
        libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
        powf = lambda x: libm.powf(float(x), 2.4)       # noqa: E731
    except OSError:                                       # pragma: no cover
        powf = lambda x: float(np.power(np.float32(x), np.float32(2.4)))   # noqa: E731
    for i in range(256):
        if not c[i] <= np.float32(0.04045):
            out[i] = np.float32(powf((c[i] + np.float32(0.055)) / np.float32(1.055)))
    return out


class _Document:
    def __init__(self, path):
        self.dir = os.path.dirname(os.path.abspath(path))
        raw = open(path, "rb").read()
        self.glb_bin = None
        if raw[:4] == b"glTF":
            magic, version, length = struct.unpack_from("<4sII", raw, 0)
            if version != 2:
                raise GltfError(f"GLB version {version} is not supported")
            off, chunks = 12, []
            while off + 8 <= min(length, len(raw)):
                clen, ctype = struct.unpack_from("<II", raw, off)
                chunks.append((ctype, raw[off + 8: off + 8 + clen]))
                off += 8 + clen + (-clen % 4)
            if not chunks or chunks[0][0] != 0x4E4F534A:
                raise GltfError("GLB without a JSON chunk")
            self.json = json.loads(chunks[0][1].decode("utf-8"))
            for ctype, data in chunks[1:]:
                if ctype == 0x004E4942:
                    self.glb_bin = data
                    break
        else:
            self.json = json.loads(raw.decode("utf-8"))
        self._buffers = {}

    def get(self, key):
        return self.json.get(key, [])

    def _uri(self, uri):
        if uri.startswith("data:"):
            return base64.b64decode(uri.split(",", 1)[1])
        return open(os.path.join(self.dir, uri), "rb").read()

    def buffer(self, i):
        if i not in self._buffers:
            b = self.get("buffers")[i]
            if "uri" in b:
                self._buffers[i] = self._uri(b["uri"])
            elif self.glb_bin is not None and i == 0:
                self._buffers[i] = self.glb_bin
            else:
                raise GltfError(f"buffer {i} has no data")
        return self._buffers[i]

    def view_bytes(self, i):
        v = self.get("bufferViews")[i]
        off = v.get("byteOffset", 0)
        return self.buffer(v["buffer"])[off: off + v["byteLength"]], v.get("byteStride", 0)

    def accessor(self, i):
        a = self.get("accessors")[i]
        if "sparse" in a:
            raise GltfError("sparse accessors are not supported")
        dt = np.dtype(_COMPONENT[a["componentType"]]).newbyteorder("<")
        nc, count = _NCOMP[a["type"]], a["count"]
        if "bufferView" not in a:
            return np.zeros((count, nc), dtype=dt), a
        data, stride = self.view_bytes(a["bufferView"])
        off = a.get("byteOffset", 0)
        elem = dt.itemsize * nc
        if stride in (0, elem):
            arr = np.frombuffer(data, dtype=dt, count=count * nc, offset=off).reshape(count, nc)
        else:
            arr = np.stack([np.frombuffer(data, dtype=dt, count=nc, offset=off + k * stride) for k in range(count)]) if count else np.zeros((0, nc), dt)
        return arr, a


def _into_f32(arr, acc):
    """`ReadTexCoords::into_f32`: floats as they are, normalised u8 / u16 divided by their maximum."""
    if arr.dtype.kind == "f":
        return arr.astype(np.float32)
    if arr.dtype == np.uint8:
        return arr.astype(np.float32) / np.float32(255.0)
    if arr.dtype == np.uint16:
        return arr.astype(np.float32) / np.float32(65535.0)
    raise GltfError(f"TEXCOORD_0 component type {acc['componentType']} is not allowed")


def _mat4_mul(a, b):
    """glam `Mat4 * Mat4` on column-major columns: col_j = ((a.x*b_j.x + a.y*b_j.y) + a.z*b_j.z) + a.w*b_j.w, in f32."""
    out = np.empty((4, 4), dtype=np.float32)          # out[:, j] = column j
    for j in range(4):
        out[:, j] = ((a[:, 0] * b[0, j] + a[:, 1] * b[1, j]) + a[:, 2] * b[2, j]) + a[:, 3] * b[3, j]
    return out


def _node_local(node):
    if "matrix" in node:
        return np.array(node["matrix"], dtype=np.float32).reshape(4, 4).T     # column-major list -> [row, col]
    t = np.array(node.get("translation", [0, 0, 0]), dtype=np.float32)
    x, y, z, w = (np.float32(v) for v in node.get("rotation", [0, 0, 0, 1]))
    s = np.array(node.get("scale", [1, 1, 1]), dtype=np.float32)
    one, two = np.float32(1), np.float32(2)
    r = np.array([[one - two * (y * y + z * z), two * (x * y - z * w), two * (x * z + y * w)],
                  [two * (x * y + z * w), one - two * (x * x + z * z), two * (y * z - x * w)],
                  [two * (x * z - y * w), two * (y * z + x * w), one - two * (x * x + y * y)]], dtype=np.float32)
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = r * s[None, :]
    m[:3, 3] = t
    return m


def _global_transforms(doc):
    nodes = doc.get("nodes")
    out = [np.eye(4, dtype=np.float32) for _ in nodes]

    def walk(i, parent):
        g = _mat4_mul(parent, _node_local(nodes[i]))
        out[i] = g
        for c in nodes[i].get("children", []):
            walk(c, g)

    for sc in doc.get("scenes"):
        for root in sc.get("nodes", []):
            walk(root, np.eye(4, dtype=np.float32))
    return out


def _decode_image(doc, img, convert_srgb, lut, notes, index):
    from PIL import Image
    if "bufferView" in img:
        data, _ = doc.view_bytes(img["bufferView"])
    elif "uri" in img:
        data = doc._uri(img["uri"])
    else:
        raise GltfError(f"image {index} has neither uri nor bufferView")
    im = Image.open(io.BytesIO(data))
    fmt = (im.format or "").upper()
    bits16 = im.mode.startswith("I;16") or getattr(im, "png", None) is not None and im.info.get("bits", 8) == 16
    im.load()
    if fmt in ("JPEG", "MPO"):
        notes.append(f"image {index}: JPEG decoded by PIL/libjpeg; texels may differ from the `image` crate's decoder by a few levels (parity unpinned)")
    if bits16:
        notes.append(f"image {index}: 16-bit PNG reduced to 8 bits by PIL (the reference keeps 16-bit RGB as value / 65535)")
    if im.mode == "P":                                  # the `image` crate expands palettes to RGB(A)8
        im = im.convert("RGBA" if "transparency" in im.info else "RGB")
    w, h = im.size
    if im.mode in ("RGB", "RGBA"):
        px = np.asarray(im, dtype=np.uint8)[..., :3]
        if convert_srgb:
            return lut[px]
        return px.astype(np.float32) / np.float32(255.0)
    if im.mode in ("RGB;16", "RGBA;16"):                # pragma: no cover  (PIL normally never reports these)
        px = np.asarray(im)[..., :3].astype(np.float32)
        return px / np.float32(65535.0)
    # L8, LA8, 16-bit grey, float: `_ => vec![Vec3::splat(0.5); w*h]` (gltf.rs:103)
    notes.append(f"image {index}: pixel format {im.mode} is not RGB(A); flat 0.5 grey like the reference")
    return np.full((h, w, 3), 0.5, dtype=np.float32)


def load_gltf(path):
    """`load_gltf` (loader/gltf.rs:17-36)."""
    doc = _Document(path)
    out = GltfScene()
    tex_to_img = [t.get("source", 0) for t in doc.get("textures")]

    def image_of(info):
        if info is None:
            return -1
        ti = info.get("index", 0)
        return tex_to_img[ti] if 0 <= ti < len(tex_to_img) else -1

    # textures: one per image; base colour / emissive images are sRGB
    srgb = set()
    for m in doc.get("materials"):
        pbr = m.get("pbrMetallicRoughness", {})
        for info in (pbr.get("baseColorTexture"), m.get("emissiveTexture")):
            if info is not None and image_of(info) >= 0:
                srgb.add(image_of(info))
    lut = _srgb_lut()
    for i, img in enumerate(doc.get("images")):
        out.textures.append(_decode_image(doc, img, i in srgb, lut, out.notes, i))

    # materials
    for m in doc.get("materials"):
        pbr = m.get("pbrMetallicRoughness", {})
        ext = m.get("extensions", {})
        out.materials.append(host.default_material(
            base_color_factor=pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0]),
            base_color_texture=image_of(pbr.get("baseColorTexture")),
            metallic_factor=pbr.get("metallicFactor", 1.0),
            roughness_factor=pbr.get("roughnessFactor", 1.0),
            metallic_roughness_texture=image_of(pbr.get("metallicRoughnessTexture")),
            normal_texture=image_of(m.get("normalTexture")),
            normal_scale=(m.get("normalTexture") or {}).get("scale", 1.0),
            emissive_texture=image_of(m.get("emissiveTexture")),
            emissive_factor=m.get("emissiveFactor", [0.0, 0.0, 0.0]),
            transmission=ext.get("KHR_materials_transmission", {}).get("transmissionFactor", 0.0),
            ior=ext.get("KHR_materials_ior", {}).get("ior", 1.5),
        ))

    # meshes and instances
    transforms = _global_transforms(doc)
    for ni, node in enumerate(doc.get("nodes")):
        if "mesh" not in node:
            continue
        for prim in doc.get("meshes")[node["mesh"]].get("primitives", []):
            if prim.get("mode", 4) != 4:
                continue
            attrs = prim.get("attributes", {})
            if "POSITION" not in attrs:
                continue
            positions = doc.accessor(attrs["POSITION"])[0].astype(np.float32)
            if len(positions) == 0:
                continue
            normals = doc.accessor(attrs["NORMAL"])[0].astype(np.float32) if "NORMAL" in attrs else np.zeros((0, 3), np.float32)
            if "TEXCOORD_0" in attrs:
                uv, acc = doc.accessor(attrs["TEXCOORD_0"])
                uvs = _into_f32(uv, acc)
            else:
                uvs = np.zeros((0, 2), np.float32)
            if "indices" in prim:
                flat = doc.accessor(prim["indices"])[0].astype(np.uint32).reshape(-1)
            else:
                flat = np.arange(len(positions), dtype=np.uint32)
            if len(flat) % 3:
                raise GltfError("index count is not a multiple of 3 (the reference would panic on the last chunk)")
            out.meshes.append(GltfMesh(positions, normals, uvs, flat.reshape(-1, 3), int(prim.get("material", 0))))
            out.instances.append((len(out.meshes) - 1, transforms[ni]))

    # the first perspective camera in node order
    for ni, node in enumerate(doc.get("nodes")):
        if "camera" not in node:
            continue
        cam = doc.get("cameras")[node["camera"]]
        if cam.get("type") != "perspective":
            continue
        t = transforms[ni]
        position = t[:3, 3].copy()
        c2 = t[:3, 2]
        inv_len = np.float32(1.0) / np.sqrt((c2[0] * c2[0] + c2[1] * c2[1]) + c2[2] * c2[2], dtype=np.float32)
        forward = -(c2 * inv_len)
        yfov = np.float32(cam["perspective"]["yfov"])
        out.camera = {"position": position, "look_at": position + forward,
                      "fov_degrees": float(yfov * np.float32(180.0 / np.pi))}
        break
    return out


def build_scene(g, sky=(0.5, 0.6, 0.8), sky_rotation=0.0, sun=None):
    """`Scene::new(blases, instances, materials, textures, camera, sky, sun)` as run_benchmark calls it
    (main.rs:104-112).  `sky`: an RGB triple (solid, the benchmark's default) or an (h, w, 3) f32 equirect image;
    `sun`: None or (direction, color)."""
    b = host.SceneBuilder()
    for t in g.textures:
        b.add_texture(t)
    for m in (g.materials or [host.default_material()]):
        b.add_material(m)
    for m in g.meshes:
        b.add_mesh(m.positions, m.indices, m.normals if len(m.normals) else None, m.uvs if len(m.uvs) else None, m.material_id)
    for blas, xf in g.instances:
        b.add_instance(blas, xf)
    sky = np.asarray(sky, dtype=np.float32)
    if sky.ndim == 1:
        b.set_sky_solid(float(sky[0]), float(sky[1]), float(sky[2]))
    else:
        b.set_sky_image(sky, sky_rotation)
    if sun is not None:
        b.set_sun(sun[0], sun[1])
    return b.build()


def camera_for(g, aspect):
    """The benchmark's camera choice (main.rs:98-102): the file's first perspective camera, else eye (0,1,3) ->
    (0,0.5,0), fov 60; aspect always replaced by the frame's."""
    if g.camera is None:
        return host.Camera.look_at((0.0, 1.0, 3.0), (0.0, 0.5, 0.0), 60.0, aspect)
    return host.Camera.look_at(g.camera["position"], g.camera["look_at"], g.camera["fov_degrees"], aspect)
