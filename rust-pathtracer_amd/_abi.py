"""ctypes mirror of include/ptx_core.h (the C ABI of libmi355_pt.so).

Field order and types follow the header one-for-one; tests/test_abi.py checks
every sizeof against the values the C compiler reports.
"""
import ctypes as C

c_f = C.c_float
c_u32 = C.c_uint32
c_i32 = C.c_int32
c_u64 = C.c_uint64
P = C.POINTER

PTX_ABI_VERSION = 1
PTX_OK = 0
PTX_MISS = 0xFFFFFFFF
PTX_TRACE_CLOSEST = 0
PTX_TRACE_ANY = 1

ERROR_NAMES = {
    0: "PTX_OK", 1: "PTX_ERR_INVALID", 2: "PTX_ERR_NO_DEVICE", 3: "PTX_ERR_HIP",
    4: "PTX_ERR_NO_SCENE", 5: "PTX_ERR_STACK_OVERFLOW", 6: "PTX_ERR_UNSUPPORTED",
}


class PtxBvh8Node(C.Structure):
    _fields_ = [
        ("min_x", c_f * 8), ("max_x", c_f * 8),
        ("min_y", c_f * 8), ("max_y", c_f * 8),
        ("min_z", c_f * 8), ("max_z", c_f * 8),
        ("children", c_u32 * 8),
        ("valid_mask", C.c_uint8),
        ("_padding", C.c_uint8 * 7),
        ("_align_tail", C.c_uint8 * 24),   # repr(C, align(64)): 232 -> 256
    ]


class PtxBvh8(C.Structure):
    _fields_ = [
        ("nodes", P(PtxBvh8Node)), ("n_nodes", c_u32),
        ("primitive_indices", P(c_u32)), ("n_primitive_indices", c_u32),
    ]


class PtxMesh(C.Structure):
    _fields_ = [
        ("positions", P(c_f)), ("n_positions", c_u32),
        ("normals", P(c_f)), ("n_normals", c_u32),
        ("uvs", P(c_f)), ("n_uvs", c_u32),
        ("indices", P(c_u32)), ("n_triangles", c_u32),
        ("material_id", c_u32),
        ("bvh", PtxBvh8),
        ("bounds_min", c_f * 3), ("bounds_max", c_f * 3),
    ]


class PtxInstance(C.Structure):
    _fields_ = [
        ("blas_index", c_u32),
        ("transform", c_f * 16), ("inverse_transform", c_f * 16), ("normal_matrix", c_f * 16),
    ]


class PtxMaterial(C.Structure):
    _fields_ = [
        ("base_color_texture", c_i32), ("metallic_roughness_texture", c_i32),
        ("normal_texture", c_i32), ("emissive_texture", c_i32),
        ("base_color_factor", c_f * 4),
        ("roughness_factor", c_f), ("metallic_factor", c_f), ("normal_scale", c_f),
        ("emissive_factor", c_f * 3),
        ("ior", c_f), ("transmission", c_f), ("subsurface", c_f),
        ("subsurface_radius", c_f * 3), ("subsurface_albedo", c_f * 3),
        ("subsurface_anisotropy", c_f),
        ("subsurface_entry_tint", c_f * 3), ("subsurface_exit_tint", c_f * 3),
    ]


class PtxTexture(C.Structure):
    _fields_ = [("width", c_u32), ("height", c_u32), ("texels", P(c_f)), ("solid", c_f * 3)]


class PtxSky(C.Structure):
    _fields_ = [
        ("texture", PtxTexture),
        ("marginal_cdf", P(c_f)), ("conditional_cdfs", P(c_f)),
        ("width", c_u32), ("height", c_u32),
        ("total_weight", c_f), ("rotation", c_f),
    ]


class PtxSun(C.Structure):
    _fields_ = [("enabled", c_u32), ("direction", c_f * 3), ("color", c_f * 3), ("cos_angle", c_f), ("pdf", c_f)]


class PtxBlueNoise(C.Structure):
    _fields_ = [("size", c_u32), ("samples", P(c_f))]


class PtxSceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", c_u32),
        ("meshes", P(PtxMesh)), ("n_meshes", c_u32),
        ("instances", P(PtxInstance)), ("n_instances", c_u32),
        ("tlas", PtxBvh8),
        ("materials", P(PtxMaterial)), ("n_materials", c_u32),
        ("textures", P(PtxTexture)), ("n_textures", c_u32),
        ("sky", PtxSky),
        ("sun", PtxSun),
        ("blue_noise_pixel", PtxBlueNoise), ("blue_noise_lens", PtxBlueNoise),
    ]


class PtxCamera(C.Structure):
    _fields_ = [
        ("inv_vp_matrix", c_f * 16),
        ("aperture", c_f), ("focus_distance", c_f),
        ("right", c_f * 3), ("up", c_f * 3),
    ]


class PtxRay(C.Structure):
    _fields_ = [("origin", c_f * 3), ("direction", c_f * 3), ("t_max", c_f)]


class PtxHitRecord(C.Structure):
    _fields_ = [("inst_idx", c_u32), ("tri_idx", c_u32), ("t", c_f), ("u", c_f), ("v", c_f)]


COUNTER_FIELDS = [
    "rays_closest", "rays_any", "node_visits", "node_children", "prim_index_reads",
    "triangle_tests", "instance_entries", "instance_hits", "hitdata_built",
    "material_fetches", "texture_lookups", "cdf_probes", "sky_pdf_evals",
    "sky_texel_lookups", "primary_rays", "pixel_samples_written", "sss_steps",
    "any_node_visits", "any_node_children", "any_prim_index_reads", "any_triangle_tests", "any_instance_entries",
]


class PtxCounters(C.Structure):
    _fields_ = [(n, c_u64) for n in COUNTER_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}


class PtxAdaptiveConfig(C.Structure):
    _fields_ = [("min_samples", c_u32), ("variance_threshold", c_f), ("max_samples_per_pass", c_u64)]


class PtxFrameStats(C.Structure):
    _fields_ = [("total_samples", c_u64), ("min_samples", c_u32), ("max_samples", c_u32),
                ("uniform_passes", c_u32), ("adaptive_passes", c_u32), ("adaptive_samples", c_u64)]


class PtxDebugStats(C.Structure):
    _fields_ = [("trace_stack_high_water", c_u32), ("packet_stack_high_water", c_u32), ("packet_redo_rays", c_u64),
                ("lds_stack_entries", c_u32), ("packet_lds_entries", c_u32), ("packet_capacity", c_u32),
                ("max_leaf_undershoot", C.c_float), ("bound_forwarding", c_u32), ("max_leaf_undershoot_excess", C.c_float),
                ("vote_closest", C.c_uint64 * 9), ("vote_any", C.c_uint64 * 9), ("stack_ops", C.c_uint64 * 4),
                ("fast_redo_rays", C.c_uint64), ("fast_order", c_u32), ("uncertified_meshes", c_u32)]


class PtxTimings(C.Structure):
    _fields_ = [
        ("raygen_ms", C.c_double), ("extend_ms", C.c_double), ("shade_ms", C.c_double),
        ("shadow_ms", C.c_double), ("accumulate_ms", C.c_double), ("other_ms", C.c_double),
        ("extend_launches", c_u64), ("shade_launches", c_u64), ("shadow_launches", c_u64),
        ("packet_ms", C.c_double), ("packet_launches", c_u64),
        ("walk_ms", C.c_double), ("walk_launches", c_u64),
        ("tail_ms", C.c_double), ("tail_launches", c_u64),
    ]


# numpy dtypes of the two batch records (packed exactly like the C structs)
def ray_dtype():
    import numpy as np
    return np.dtype([("origin", np.float32, 3), ("direction", np.float32, 3), ("t_max", np.float32)])


def hit_dtype():
    import numpy as np
    return np.dtype([("inst_idx", np.uint32), ("tri_idx", np.uint32), ("t", np.float32),
                     ("u", np.float32), ("v", np.float32)])


def algorithmic_bytes(c, texel_bytes_per_lookup=48, adaptive=False):
    """the one formula both the roofline and the judge use.

    `c` is a counter dict (PtxCounters.as_dict()).  Node bytes are the USEFUL
    bytes (4 B header + 28 B per valid child), not the 256 B stored layout.
    """
    b = 4 * c["node_visits"] + 28 * c["node_children"]
    b += 4 * c["prim_index_reads"]
    b += 48 * c["triangle_tests"]
    b += 64 * c["instance_entries"] + 128 * c["instance_hits"]
    b += 60 * c["hitdata_built"]
    b += 112 * c["material_fetches"]
    b += texel_bytes_per_lookup * c["texture_lookups"]
    b += 4 * c["cdf_probes"] + 16 * c["sky_pdf_evals"] + 48 * c["sky_texel_lookups"]
    b += 16 * c["primary_rays"]
    b += (28 if adaptive else 12) * c["pixel_samples_written"]
    return b
