// pt_types.h — HBM layouts of the render core (see DESIGN.md "Data layout").
#pragma once
#include "pt_vec.h"
#include "ptx_core.h"

namespace ptd {

// ---- BVH node record: one or two 128-B units (one L2 line each), up to 8 children ------------
// The reference's Bvh8Node stores 8 full-precision child boxes SoA in 256 B
// (accel/bvh.rs:86-98), of which its own builder only ever fills 1-3
// (SURVEY.md F8b).  A device record keeps the valid slots with the same f32
// values, child k as two float4
//   c[2k]   = (min.x, min.y, min.z, max.x)
//   c[2k+1] = (max.y, max.z, child_word, meta)
// in VISIT ORDER: the reference tests the leaf slots of a node at once, in ascending slot order, and pushes the
// internal slots, which its LIFO stack then pops in descending slot order (accel/bvh.rs:339-363) — a depth-first walk
// that takes a node's children in the order [leaf slots ascending, internal slots descending].  Records list their
// children in exactly that order, so the traversal is one rule: push the hit children in reverse list order, pop.
// A node with <= 4 children is one unit, with 5-8 two consecutive units (children 4-7 in the second).
// child_word: bit31 = leaf; leaf = 1<<31 | (count-1)<<24 | prim_start (start relative to the owning BVH's leaf array,
// as in the reference); internal = (children_of_target_record - 1) << 28 | ABSOLUTE unit index in SceneDev::nodes —
// the parent says how many children the child record holds.
// meta (child 0 only, counters): bits 0-3 = popcount(valid_mask) of the reference node, bit 4 = this record is a
// reference node (set in the faithful tree, clear in the wide tree).
//
// Every BVH is stored in two forms.  The FAITHFUL tree has one record per reference node.  A WIDE tree is an
// order-preserving collapse of it: an internal child is replaced, in place, by that child's own list, largest
// surface area first, until a record holds 8 children.  Substituting a list for its parent keeps the depth-first
// order of the LEAVES unchanged, and that order is all the result depends on: in the reference a leaf's primitives
// are tested iff the ray hits the leaf's box and `t_enter < closest_t` at the moment the walk reaches it — the tests
// of its ancestors are implied, because child boxes are exact min/max unions (an ancestor's slab interval contains
// the leaf's, also after f32 rounding, which is monotone) and closest_t only shrinks.  That argument needs ordered
// comparisons, so rays that can produce NaN slabs (a zero direction component or a non-finite origin: 0 * inf) walk
// the faithful tree, as do the counted kernels, whose event counts must be the reference's.  The packet kernels walk
// an 8-wide tree (one record fetch serves the wave's 64 rays); the per-ray kernels walk the faithful tree, because they
// are VALU-issue bound and wider records mean more box tests per ray (measured: DESIGN.md).
struct NodeRec { float4 c[8]; };
static_assert(sizeof(NodeRec) == 128, "node record unit must be one 128-byte line");

constexpr uint32_t LEAF_FLAG = 0x80000000u;
// Shape certificate (upload): a mesh with a triangle whose edges at v0 are nearly parallel is UNCERTIFIED — the shortcuts
// of the closest-hit kernels (forwarded bound, nearest-first order) presume triangle hits that sit inside the boxes above
// them, which such triangles break.  Marked in two places: bit 31 of the per-ray root word in the instance record
// (InstRec::m[4].z; a mesh root is never a leaf word), and META_UNCERT in the meta word of every TLAS child whose subtree
// holds an instance of such a mesh (children 1.. of a record use their meta word for nothing else).
constexpr uint32_t INST_UNCERT = 0x80000000u;
constexpr uint32_t META_UNCERT = 1u << 8;
constexpr uint32_t NODE_NCHILD_SHIFT = 28;
constexpr uint32_t NODE_NCHILD_MASK = 7u;
constexpr uint32_t NODE_INDEX_MASK = (1u << NODE_NCHILD_SHIFT) - 1u;
constexpr uint32_t PRIM_START_MASK = 0x00FFFFFFu;
constexpr uint32_t PRIM_COUNT_SHIFT = 24;
constexpr uint32_t PRIM_COUNT_MASK = 0x7Fu;

// ---- leaf triangle record: 48 B, in primitive_indices order -------------------
//   a = (v0.x, v0.y, v0.z, e1.x)  b = (e1.y, e1.z, e2.x, e2.y)  c = (e2.z, tri_idx, -, -)
// e1 = v1 - v0, e2 = v2 - v0 are the first two operations of intersect_triangle
// (shape/triangle_mesh.rs:70-71) done once at upload: same inputs, same
// rounding, and two index indirections fewer per test.
struct TriRec { float4 a, b, c; };
static_assert(sizeof(TriRec) == 48, "triangle record");

// ---- instance record: 128 B ---------------------------------------------------
//   m[0..2] inverse_transform columns packed (x.xyz,y.x | y.yz,z.xy | z.z,w.xyz)
//   m[3]    (bounds_min.xyz, bounds_max.x)      -- the Blas-level item box
//   m[4]    (bounds_max.y, bounds_max.z, root word of the mesh BVH in the per-ray kernels' tree, tri_base)   (other trees: SceneDev::inst_root_*)
//   m[5..7] transform columns packed like m[0..2] (read only on an accepted hit)
struct InstRec { float4 m[8]; };
static_assert(sizeof(InstRec) == 128, "instance record");

// normal_matrix 3x3 part (x.xyz, y.xyz, z.xyz) + mesh id: read once per path vertex
// shade_base = first ShadeTri of the instance's mesh; mat_flags = material id | SHADE_HAS_NORMALS | SHADE_HAS_UVS, so a
// path vertex reaches its shading data in two dependent reads (InstShade -> ShadeTri) instead of four
struct InstShade { float n[9]; uint32_t mesh_id; uint32_t shade_base; uint32_t mat_flags; };
constexpr uint32_t SHADE_HAS_NORMALS = 0x80000000u, SHADE_HAS_UVS = 0x40000000u, SHADE_MATERIAL_MASK = 0x3FFFFFFFu;

// ---- shading triangle: 80 B, by the mesh's own triangle index ----------------------------
//   a = (face_normal.xyz, n0.x)  b = (n0.yz, n1.xy)  c = (n1.z, n2.xyz)  d = (uv0, uv1)  e = (uv2, -, -)
// Everything compute_hit_data (shape/triangle_mesh.rs:102-190) gathers per hit — index triple, three positions, three
// normals, three uvs: ten scattered 64-B sectors — packed at upload into one record (two sectors).  face_normal =
// normalize(cross(v1 - v0, v2 - v0)) is evaluated once per triangle by a device kernel with the very code a path
// vertex used to run, so the bits are the same.
struct ShadeTri { float4 a, b, c, d, e; };
static_assert(sizeof(ShadeTri) == 80, "shading triangle record");

struct MeshRec {
    uint32_t pos_base;      // float index into positions
    uint32_t nrm_base;      // float index into normals, 0xFFFFFFFF = none
    uint32_t uv_base;       // float index into uvs, 0xFFFFFFFF = none
    uint32_t idx_base;      // u32 index into indices
    uint32_t material_id;
    uint32_t node_base, tri_base, n_triangles;
};

struct TexRec {
    uint32_t width, height;
    uint64_t offset;        // float index into the texel pool; ~0ull = SolidColor
    float solid[3];
    uint32_t pad;
};

struct SceneDev {
    const NodeRec* nodes;          // TLAS records first, then every mesh BVH
    const uint32_t* tlas_prims;    // TLAS primitive_indices (instance ids)
    const TriRec* tris;
    const InstRec* inst;           // by instance id
    const InstRec* inst_tlas;      // the same records in TLAS primitive_indices order (one load fewer per entry)
    const uint32_t* inst_root_faithful;   // faithful-tree root word of each entry of inst_tlas (NaN-slab rays, counted kernels)
    const uint32_t* inst_root_packet;     // wide-tree root word the packet kernels enter the instance's mesh BVH through
    const InstShade* inst_shade;
    const MeshRec* meshes;
    const ShadeTri* shade_tris;
    const float* positions;
    const float* normals;
    const float* uvs;
    const uint32_t* indices;
    const PtxMaterial* materials;
    const TexRec* textures;
    const float* texels;
    // sky
    TexRec sky_tex;
    const float* sky_marginal;
    const float* sky_conditional;
    // 16-ary search tables over the two CDFs (built at upload when both are non-decreasing; see cdf_count16):
    // level 0 = the CDF without its last entry, padded with +inf to a multiple of 16 (at least one pad); level l+1 =
    // every 16th entry (the group maxima) of level l, padded the same way; the top level is one group.
    const float* sky_search;           // all levels of the marginal, then all levels of every conditional row
    uint32_t cdf_levels_m, cdf_levels_c;   // 0 = tables absent: binary search
    uint32_t cdf_off_m[4];                 // float offset of marginal level l
    uint32_t cdf_off_c[4], cdf_stride_c[4];   // conditional level l: offset of row 0, floats per row
    uint32_t sky_w, sky_h;
    float sky_total_weight, sky_rotation;
    // sun
    uint32_t sun_enabled;
    float sun_dir[3], sun_color[3], sun_cos_angle, sun_pdf;
    // blue noise (64x64 Vec2 each)
    const float* bn_pixel;
    const float* bn_lens;
    uint32_t bn_size;
    uint32_t n_tlas_nodes;         // 0 = empty TLAS (Tlas::build with no instances)
    uint32_t tlas_root_word;       // faithful TLAS: record 0 | (children - 1) << 28
    uint32_t tlas_root_word_ray;   // TLAS the per-ray kernels walk (== tlas_root_word unless PTX_WIDE_RAY_BVH asks for a collapse)
    uint32_t tlas_root_word_packet;   // wide TLAS of the packet kernels (== tlas_root_word when the collapse is off)
    uint32_t has_sss;
    // forwarded_bound (pt_traverse.h): rounding bound of world -> local -> world, certified per scene at upload
    uint32_t fwd_on;
    float fwd_lin, fwd_t;
    // FAST traversal (pt_traverse.h): rays whose origin has |o|_1 <= fast_omax (the scene's extent) may take it;
    // fast_slack = the world/local rounding bound at that extent, the absolute part of every FAST margin
    uint32_t fast_on;
    float fast_omax, fast_slack;                           // (the relative margins are compile-time constants: pt_traverse.h FAST_*)
};

struct CameraDev {
    float inv_vp[16];
    float aperture, focus_distance;
    float right[3], up[3];
};

// Per-wave event counters behind ptx_get_counters.
struct DevCounters {
    unsigned long long v[sizeof(PtxCounters) / 8];
    // PtxDebugStats (not part of PtxCounters): [0] deepest per-ray stack (counted kernels), [1] deepest packet stack,
    // [2..3] rays handed from overflowed packets to k_extend_redo (u64), [4] largest leaf undershoot (f32 bits, counted kernels)
    unsigned int dbg[8];      // ... [6..7] rays the FAST closest-hit kernel handed to the reference-order kernel (u64)
    // the wave vote of the counted per-ray kernels (LaneCnt::vote), closest-hit [0..8] then any-hit [9..17]
    unsigned long long vote[18];
    unsigned long long stk[4];     // counted per-ray kernels: node-visit stack entries read, read from the HBM strips, written, written to the strips
};
enum CounterId {
    C_RAYS_CLOSEST, C_RAYS_ANY, C_NODE_VISITS, C_NODE_CHILDREN, C_PRIM_INDEX_READS, C_TRI_TESTS,
    C_INST_ENTRIES, C_INST_HITS, C_HITDATA, C_MATERIAL, C_TEXTURE, C_CDF_PROBES, C_SKY_PDF,
    C_SKY_TEXEL, C_PRIMARY, C_PIXEL_SAMPLES, C_SSS_STEPS,
    C_ANY_NODE_VISITS, C_ANY_NODE_CHILDREN, C_ANY_PRIM_INDEX_READS, C_ANY_TRI_TESTS, C_ANY_INST_ENTRIES, C_COUNT
};
static_assert(C_COUNT == sizeof(PtxCounters) / 8, "counter ids must mirror PtxCounters");

}  // namespace ptd
