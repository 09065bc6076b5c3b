"""Host side of the renderer (ctypes over libpt_host.so).

Mirrors the part of the reference that stays on the host — `Scene::new`
(scene.rs:22-40), `TriangleMesh::new` + `Blas::build` (loader/gltf.rs:300-302),
`Camera::new` (camera.rs:16-33), `Material` (material.rs), `DirectionalLight::new`
(light.rs:13-15) — and hands the flattened `PtxSceneDesc` to the render core.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as abi

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libpt_host.so")

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HOST_LIB_PATH):
        raise RuntimeError(f"{HOST_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(HOST_LIB_PATH)
    vp = C.c_void_p
    L.pth_builder_new.restype = vp
    L.pth_builder_free.argtypes = [vp]
    L.pth_builder_add_mesh.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32]
    L.pth_builder_add_mesh.restype = C.c_int32
    L.pth_builder_add_instance.argtypes = [vp, C.c_uint32, vp]
    L.pth_builder_add_instance.restype = C.c_int32
    L.pth_builder_add_material.argtypes = [vp, C.POINTER(abi.PtxMaterial)]
    L.pth_builder_add_material.restype = C.c_int32
    L.pth_builder_add_texture.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.pth_builder_add_texture.restype = C.c_int32
    L.pth_builder_set_sky_solid.argtypes = [vp, C.c_float, C.c_float, C.c_float]
    L.pth_builder_set_sky_image.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.c_float]
    L.pth_builder_set_sun.argtypes = [vp, vp, vp]
    L.pth_builder_finish.argtypes = [vp]
    L.pth_builder_finish.restype = vp
    L.pth_scene_desc.argtypes = [vp]
    L.pth_scene_desc.restype = C.POINTER(abi.PtxSceneDesc)
    L.pth_scene_free.argtypes = [vp]
    L.pth_scene_stats.argtypes = [vp, C.POINTER(C.c_uint64 * 4)]
    L.pth_material_default.argtypes = [C.POINTER(abi.PtxMaterial)]
    L.pth_camera_new.argtypes = [vp, vp, C.c_float, C.c_float, C.POINTER(abi.PtxCamera)]
    L.pth_camera_from_angles.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(abi.PtxCamera)]
    L.pth_bluenoise_generate.argtypes = [C.c_uint32, C.c_uint64, vp]
    L.pth_bvh8_build.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.pth_bvh8_build.restype = vp
    L.pth_bvh8_view.argtypes = [vp, C.POINTER(abi.PtxBvh8)]
    L.pth_bvh8_free.argtypes = [vp]
    L.pth_scene_procedural.argtypes = [C.c_char_p, C.c_uint64, C.c_float, C.c_uint32]
    L.pth_scene_procedural.restype = vp
    L.pth_scene_default_camera.argtypes = [C.c_char_p, C.c_float, C.POINTER(abi.PtxCamera)]
    L.pth_scene_default_camera.restype = C.c_int32
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def default_material(**overrides):
    """`Material::default()` with fields overridden (material.rs:38-65)."""
    m = abi.PtxMaterial()
    lib().pth_material_default(C.byref(m))
    for k, v in overrides.items():
        cur = getattr(m, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(m, k, v)
    return m


FLAG_APP_LIGHTING = 1   # procedural HDR sky + the app's sun (app.rs:121-132) instead of --benchmark's solid sky
FLAG_SSS_FRACTION = 2   # ~10 % of materials get subsurface = 1 (BASELINE.json config 5)


class Scene:
    """An immutable built scene (what `Scene::new` returns) + its C descriptor."""

    def __init__(self, handle, name="custom"):
        if not handle:
            raise ValueError("scene construction failed")
        self._h = handle
        self.name = name
        self.desc = lib().pth_scene_desc(handle)

    @classmethod
    def procedural(cls, name, seed=1, detail=1.0, flags=0):
        h = lib().pth_scene_procedural(name.encode(), int(seed), float(detail), int(flags))
        if not h:
            raise ValueError(f"unknown procedural scene {name!r}")
        return cls(h, name)

    def stats(self):
        out = (C.c_uint64 * 4)()
        lib().pth_scene_stats(self._h, C.byref(out))
        return {"triangles": int(out[0]), "bvh_nodes": int(out[1]), "instances": int(out[2]), "texel_bytes": int(out[3])}

    def default_camera(self, aspect):
        cam = abi.PtxCamera()
        if lib().pth_scene_default_camera(self.name.encode(), float(aspect), C.byref(cam)) != 0:
            return Camera.look_at((0.0, 1.0, 3.0), (0.0, 0.5, 0.0), 60.0, aspect)   # main.rs:98-100
        return cam

    def __del__(self):
        try:
            if self._h:
                lib().pth_scene_free(self._h)
                self._h = None
        except Exception:
            pass


class SceneBuilder:
    """Collects what the reference's loader collects, then `build()` == `Scene::new`."""

    def __init__(self):
        self._b = lib().pth_builder_new()

    def add_mesh(self, positions, indices, normals=None, uvs=None, material_id=0):
        p = _f32(positions).reshape(-1, 3)
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        n = _f32(normals).reshape(-1, 3) if normals is not None else None
        u = _f32(uvs).reshape(-1, 2) if uvs is not None else None
        r = lib().pth_builder_add_mesh(self._b, _ptr(p), len(p), _ptr(n), 0 if n is None else len(n),
                                       _ptr(u), 0 if u is None else len(u), _ptr(idx), len(idx), material_id)
        if r < 0:
            raise ValueError("invalid mesh")
        return r

    def add_instance(self, blas_index, transform=None):
        m = _f32(np.eye(4) if transform is None else transform).reshape(4, 4)
        cols = np.ascontiguousarray(m.T).reshape(16)   # row-major numpy -> column-major glam
        r = lib().pth_builder_add_instance(self._b, blas_index, _ptr(cols))
        if r < 0:
            raise ValueError("invalid instance")
        return r

    def add_material(self, material):
        return lib().pth_builder_add_material(self._b, C.byref(material))

    def add_texture(self, texels):
        t = _f32(texels)
        h, w = t.shape[0], t.shape[1]
        return lib().pth_builder_add_texture(self._b, w, h, _ptr(t))

    def set_sky_solid(self, r, g, b):
        lib().pth_builder_set_sky_solid(self._b, r, g, b)

    def set_sky_image(self, texels, rotation=0.0):
        t = _f32(texels)
        lib().pth_builder_set_sky_image(self._b, t.shape[1], t.shape[0], _ptr(t), rotation)

    def set_sun(self, direction, color):
        d, c = _f32(direction), _f32(color)
        lib().pth_builder_set_sun(self._b, _ptr(d), _ptr(c))

    def build(self):
        return Scene(lib().pth_builder_finish(self._b))

    def __del__(self):
        try:
            lib().pth_builder_free(self._b)
        except Exception:
            pass


class Camera:
    @staticmethod
    def look_at(position, look_at, fov_degrees, aspect):
        """`Camera::new` + `update_matrix` (camera.rs:16-60)."""
        cam = abi.PtxCamera()
        p, l = _f32(position), _f32(look_at)
        lib().pth_camera_new(_ptr(p), _ptr(l), fov_degrees, aspect, C.byref(cam))
        return cam

    @staticmethod
    def from_angles(position, yaw, pitch, fov_degrees, aspect, aperture=0.0, focus_distance=10.0):
        cam = abi.PtxCamera()
        p = _f32(position)
        lib().pth_camera_from_angles(_ptr(p), yaw, pitch, fov_degrees, aspect, aperture, focus_distance, C.byref(cam))
        return cam


def bluenoise_generate(size, seed):
    """`BlueNoiseTexture::generate` (bluenoise.rs:9-20) -> (size*size, 2) f32."""
    out = np.zeros((size * size, 2), dtype=np.float32)
    lib().pth_bluenoise_generate(size, seed, _ptr(out))
    return out


class Bvh8:
    """`Bvh8::build(primitive_bounds, max_prims_per_leaf)` on its own (accel/bvh.rs:279-287)."""

    def __init__(self, bounds, max_prims_per_leaf):
        b = _f32(bounds).reshape(-1, 6)
        self._h = lib().pth_bvh8_build(_ptr(b), len(b), max_prims_per_leaf)
        self.view = abi.PtxBvh8()
        lib().pth_bvh8_view(self._h, C.byref(self.view))

    def nodes(self):
        n = self.view.n_nodes
        if n == 0:
            return np.zeros((0, 64), dtype=np.uint32)
        raw = np.ctypeslib.as_array(C.cast(self.view.nodes, C.POINTER(C.c_uint32)), shape=(n, 64)).copy()
        return raw

    def primitive_indices(self):
        n = self.view.n_primitive_indices
        if n == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(self.view.primitive_indices, shape=(n,)).copy()

    def __del__(self):
        try:
            lib().pth_bvh8_free(self._h)
        except Exception:
            pass
