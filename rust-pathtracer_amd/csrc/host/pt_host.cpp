// pt_host.cpp — host-side scene construction (see include/pt_host.h).
// Follows the reference's build half so that node contents and slot order —
// which decide traversal tie-breaks — are the ones the Rust
// host would hand over.  Citations are into /root/reference/src/raytracer/.
#include "pt_host_internal.h"

#include <algorithm>
#include <mutex>

namespace pth {

// ---------------------------------------------------------------- Aabb ------
// accel/bvh.rs:10-84
static inline void grow(Aabb& a, const Aabb& o) { a.mn = vmin(a.mn, o.mn); a.mx = vmax(a.mx, o.mx); }
static inline void grow_point(Aabb& a, V3 p) { a.mn = vmin(a.mn, p); a.mx = vmax(a.mx, p); }
static inline float surface_area(const Aabb& a) {
    const V3 d = a.mx - a.mn;
    return 2.0f * ((d.x * d.y + d.y * d.z) + d.z * d.x);
}
static inline V3 center(const Aabb& a) { return (a.mn + a.mx) * 0.5f; }

// ---------------------------------------------------------------- Bvh2 ------
// accel/bvh.rs:418-691 — recursive binned SAH, 16 bins, costs 1/1.
namespace {
constexpr int SAH_BINS = 16;
constexpr float TRAVERSAL_COST = 1.0f;
constexpr float INTERSECTION_COST = 1.0f;

struct PrimitiveInfo { uint32_t index; V3 centroid; Aabb bounds; };
struct Bvh2Node { Aabb aabb; uint32_t right_or_prim; uint32_t count; bool is_leaf() const { return count > 0; } };
struct Bvh2 { std::vector<Bvh2Node> nodes; std::vector<uint32_t> primitive_indices; };
struct Bin { Aabb bounds = Aabb::empty(); uint32_t count = 0; };

struct Split { int axis; float pos; float cost; };

Split find_best_split(const PrimitiveInfo* prims, size_t n, const Aabb& centroid_bounds, const Aabb& node_bounds) {
    const float node_area = surface_area(node_bounds);
    if (node_area <= 0.0f) return Split{0, 0.0f, INF};

    const V3 extent = centroid_bounds.mx - centroid_bounds.mn;
    Split best{0, 0.0f, INF};

    for (int axis = 0; axis < 3; ++axis) {
        if (extent[axis] < 1e-7f) continue;

        const float scale = (float)SAH_BINS / extent[axis];
        const float base = centroid_bounds.mn[axis];
        Bin bins[SAH_BINS];

        for (size_t i = 0; i < n; ++i) {
            float b = rs_clamp((prims[i].centroid[axis] - base) * scale, 0.0f, (float)(SAH_BINS - 1));
            size_t bin = (b != b) ? 0 : (size_t)b;  // `as usize`: NaN -> 0
            grow(bins[bin].bounds, prims[i].bounds);
            bins[bin].count += 1;
        }

        Aabb left_bounds[SAH_BINS - 1];
        uint32_t left_counts[SAH_BINS - 1];
        Aabb run_b = Aabb::empty();
        uint32_t run_c = 0;
        for (int i = 0; i < SAH_BINS - 1; ++i) {
            grow(run_b, bins[i].bounds);
            run_c += bins[i].count;
            left_bounds[i] = run_b;
            left_counts[i] = run_c;
        }

        run_b = Aabb::empty();
        run_c = 0;
        for (int i = SAH_BINS - 2; i >= 0; --i) {
            grow(run_b, bins[i + 1].bounds);
            run_c += bins[i + 1].count;
            if (left_counts[i] == 0 || run_c == 0) continue;

            const float cost = TRAVERSAL_COST +
                INTERSECTION_COST * ((float)left_counts[i] * surface_area(left_bounds[i]) +
                                     (float)run_c * surface_area(run_b)) / node_area;
            if (cost < best.cost) {
                best = Split{axis, centroid_bounds.mn[axis] + (float)(i + 1) * extent[axis] / (float)SAH_BINS, cost};
            }
        }
    }
    return best;
}

// accel/bvh.rs:677-691 — swap-from-the-right partition (unstable on purpose).
size_t partition(PrimitiveInfo* p, size_t n, int axis, float split_pos) {
    size_t left = 0, right = n;
    while (left < right) {
        if (p[left].centroid[axis] < split_pos) {
            left += 1;
        } else {
            right -= 1;
            std::swap(p[left], p[right]);
        }
    }
    return left;
}

uint32_t create_leaf(Bvh2& out, const PrimitiveInfo* prims, size_t n, const Aabb& bounds) {
    const uint32_t node_idx = (uint32_t)out.nodes.size();
    const uint32_t first_prim = (uint32_t)out.primitive_indices.size();
    for (size_t i = 0; i < n; ++i) out.primitive_indices.push_back(prims[i].index);
    out.nodes.push_back(Bvh2Node{bounds, first_prim, (uint32_t)n});
    return node_idx;
}

uint32_t build_bvh2_recursive(std::vector<PrimitiveInfo>& primitives, size_t start, size_t end, Bvh2& out,
                              uint32_t max_prims_per_leaf) {
    const uint32_t node_idx = (uint32_t)out.nodes.size();
    const size_t count = end - start;
    PrimitiveInfo* prims = primitives.data() + start;

    Aabb bounds = Aabb::empty();
    for (size_t i = 0; i < count; ++i) grow(bounds, prims[i].bounds);

    if ((uint32_t)count <= max_prims_per_leaf) return create_leaf(out, prims, count, bounds);

    Aabb centroid_bounds = Aabb::empty();
    for (size_t i = 0; i < count; ++i) grow_point(centroid_bounds, prims[i].centroid);

    const V3 extent = centroid_bounds.mx - centroid_bounds.mn;
    if (rs_max(rs_max(extent.x, extent.y), extent.z) < 1e-7f) return create_leaf(out, prims, count, bounds);

    const Split s = find_best_split(prims, count, centroid_bounds, bounds);
    const float leaf_cost = INTERSECTION_COST * (float)count;
    if (s.cost >= leaf_cost) return create_leaf(out, prims, count, bounds);

    size_t mid = start + partition(prims, count, s.axis, s.pos);
    if (mid == start || mid == end) {
        const int axis = s.axis;
        std::stable_sort(prims, prims + count, [axis](const PrimitiveInfo& a, const PrimitiveInfo& b) {
            return a.centroid[axis] < b.centroid[axis];
        });
        mid = start + count / 2;
    }

    out.nodes.push_back(Bvh2Node{bounds, 0, 0});
    build_bvh2_recursive(primitives, start, mid, out, max_prims_per_leaf);
    const uint32_t right_idx = (uint32_t)out.nodes.size();
    out.nodes[node_idx].right_or_prim = right_idx;
    build_bvh2_recursive(primitives, mid, end, out, max_prims_per_leaf);
    return node_idx;
}

Bvh2 bvh2_build(const std::vector<std::pair<uint32_t, Aabb>>& primitive_bounds, uint32_t max_prims_per_leaf) {
    std::vector<PrimitiveInfo> primitives;
    primitives.reserve(primitive_bounds.size());
    for (const auto& pb : primitive_bounds) primitives.push_back(PrimitiveInfo{pb.first, center(pb.second), pb.second});
    Bvh2 out;
    if (primitives.empty()) {
        out.nodes.push_back(Bvh2Node{Aabb::empty(), 0, 0});
        return out;
    }
    out.nodes.reserve(2 * primitives.size());
    out.primitive_indices.reserve(primitives.size());
    build_bvh2_recursive(primitives, 0, primitives.size(), out, max_prims_per_leaf);
    return out;
}

// ---------------------------------------------------------------- Bvh8 ------
// accel/bvh.rs:86-169 (node + child codec) and :693-814 (collapse).
PtxBvh8Node default_node() {
    PtxBvh8Node n;
    std::memset(&n, 0, sizeof(n));
    for (int i = 0; i < 8; ++i) {
        n.min_x[i] = n.min_y[i] = n.min_z[i] = INF;
        n.max_x[i] = n.max_y[i] = n.max_z[i] = -INF;
    }
    return n;
}
void set_child_aabb(PtxBvh8Node& n, size_t slot, const Aabb& a) {
    n.min_x[slot] = a.mn.x; n.max_x[slot] = a.mx.x;
    n.min_y[slot] = a.mn.y; n.max_y[slot] = a.mx.y;
    n.min_z[slot] = a.mn.z; n.max_z[slot] = a.mx.z;
}
void set_child_internal(PtxBvh8Node& n, size_t slot, uint32_t node_idx) {
    n.children[slot] = node_idx;
    n.valid_mask |= (uint8_t)(1u << slot);
}
void set_child_leaf(PtxBvh8Node& n, size_t slot, uint32_t prim_start, uint32_t prim_count) {
    // release-mode arithmetic of the reference: no range check (debug_assert only)
    n.children[slot] = 0x80000000u | ((prim_count - 1u) << 24) | prim_start;
    n.valid_mask |= (uint8_t)(1u << slot);
}

struct Bvh8Builder {
    const Bvh2& bvh2;
    Bvh8Data& out;

    // accel/bvh.rs:757-813.  The loop body clears `frontier` on every path
    // that reaches its end, so it runs once: the two BVH2 children, then one
    // expansion of the largest-area internal child.
    void collect_children(size_t bvh2_root, std::vector<size_t>& children, size_t max_children) const {
        const Bvh2Node& root_node = bvh2.nodes[bvh2_root];
        if (root_node.is_leaf()) return;

        std::vector<size_t> frontier{bvh2_root + 1, (size_t)root_node.right_or_prim};
        while (!frontier.empty() && children.size() + frontier.size() <= max_children) {
            children.insert(children.end(), frontier.begin(), frontier.end());
            if (children.size() >= max_children) break;

            bool have_best = false;
            size_t best_idx = 0;
            float best_area = 0.0f;
            for (size_t i = 0; i < children.size(); ++i) {
                const Bvh2Node& c = bvh2.nodes[children[i]];
                if (!c.is_leaf()) {
                    const float area = surface_area(c.aabb);
                    if (area > best_area) { best_area = area; best_idx = i; have_best = true; }
                }
            }

            if (have_best) {
                if (children.size() < max_children) {
                    const size_t child_idx = children[best_idx];
                    const size_t left = child_idx + 1;
                    const size_t right = bvh2.nodes[child_idx].right_or_prim;
                    children.erase(children.begin() + (ptrdiff_t)best_idx);
                    children.push_back(left);
                    if (children.size() < max_children) {
                        children.push_back(right);
                    } else {
                        frontier.assign(1, right);
                        continue;
                    }
                }
            } else {
                break;
            }
            frontier.clear();
        }
    }

    uint32_t build_recursive(size_t bvh2_root) {
        const Bvh2Node& bvh2_node = bvh2.nodes[bvh2_root];
        std::vector<size_t> children;
        children.reserve(8);
        collect_children(bvh2_root, children, 8);

        if (children.empty()) {
            // whole BVH is one leaf (accel/bvh.rs:706-723)
            const uint32_t node_idx = (uint32_t)out.nodes.size();
            PtxBvh8Node node = default_node();
            const uint32_t prim_start = (uint32_t)out.primitive_indices.size();
            const size_t first = bvh2_node.right_or_prim, count = bvh2_node.count;
            out.primitive_indices.insert(out.primitive_indices.end(), bvh2.primitive_indices.begin() + (ptrdiff_t)first,
                                         bvh2.primitive_indices.begin() + (ptrdiff_t)(first + count));
            set_child_aabb(node, 0, bvh2_node.aabb);
            set_child_leaf(node, 0, prim_start, (uint32_t)count);
            out.nodes.push_back(node);
            return node_idx;
        }

        const size_t node_idx = out.nodes.size();
        out.nodes.push_back(default_node());

        for (size_t slot = 0; slot < children.size(); ++slot) {
            const size_t bvh2_idx = children[slot];
            const Bvh2Node child_node = bvh2.nodes[bvh2_idx];
            set_child_aabb(out.nodes[node_idx], slot, child_node.aabb);
            if (child_node.is_leaf()) {
                const uint32_t prim_start = (uint32_t)out.primitive_indices.size();
                const size_t first = child_node.right_or_prim, count = child_node.count;
                out.primitive_indices.insert(out.primitive_indices.end(),
                                             bvh2.primitive_indices.begin() + (ptrdiff_t)first,
                                             bvh2.primitive_indices.begin() + (ptrdiff_t)(first + count));
                set_child_leaf(out.nodes[node_idx], slot, prim_start, (uint32_t)count);
            } else {
                const uint32_t child_idx = build_recursive(bvh2_idx);
                set_child_internal(out.nodes[node_idx], slot, child_idx);
            }
        }
        return (uint32_t)node_idx;
    }
};
}  // namespace

// accel/bvh.rs:279-306
Bvh8Data bvh8_build(const std::vector<std::pair<uint32_t, Aabb>>& primitive_bounds, uint32_t max_prims_per_leaf) {
    Bvh8Data out;
    if (primitive_bounds.empty()) return out;
    const Bvh2 bvh2 = bvh2_build(primitive_bounds, max_prims_per_leaf);
    if (bvh2.nodes.empty() || bvh2.primitive_indices.empty()) return out;
    out.nodes.reserve(bvh2.nodes.size() / 2 + 1);
    out.primitive_indices.reserve(bvh2.primitive_indices.size());
    Bvh8Builder builder{bvh2, out};
    builder.build_recursive(0);
    return out;
}

// ----------------------------------------------------------- TriangleMesh ---
// shape/triangle_mesh.rs:17-61
void MeshData::build() {
    bounds = Aabb::empty();
    for (size_t i = 0; i + 2 < positions.size(); i += 3) grow_point(bounds, v3(positions[i], positions[i + 1], positions[i + 2]));

    const float AABB_EPSILON = 1e-5f;
    const size_t ntri = indices.size() / 3;
    std::vector<std::pair<uint32_t, Aabb>> tri_bounds;
    tri_bounds.reserve(ntri);
    auto pos = [&](uint32_t i) { return v3(positions[3 * (size_t)i], positions[3 * (size_t)i + 1], positions[3 * (size_t)i + 2]); };
    for (size_t t = 0; t < ntri; ++t) {
        const V3 v0 = pos(indices[3 * t]), v1 = pos(indices[3 * t + 1]), v2 = pos(indices[3 * t + 2]);
        Aabb a{v0, v0};
        grow_point(a, v1);
        grow_point(a, v2);
        a.mn = a.mn - splat(AABB_EPSILON);
        a.mx = a.mx + splat(AABB_EPSILON);
        tri_bounds.emplace_back((uint32_t)t, a);
    }
    bvh = bvh8_build(tri_bounds, 4);
}

// ------------------------------------------------------------- Texture ------
// texture.rs:64-101
static inline float rem_euclid1(float x) {
    // f32::rem_euclid(1.0): r = x % 1.0; if r < 0 { r + 1.0 } else { r }
    float r = std::fmod(x, 1.0f);
    return (r < 0.0f) ? r + 1.0f : r;
}
static inline int rem_euclid_i(int a, int b) {
    int r = a % b;
    return r < 0 ? r + b : r;
}
V3 TextureData::sample(float uvx, float uvy) const {
    if (texels.empty()) return solid;
    const float u = rem_euclid1(uvx), v = rem_euclid1(uvy);
    const float px = u * (float)width - 0.5f;
    const float py = v * (float)height - 0.5f;
    const int x0 = (int)std::floor(px), y0 = (int)std::floor(py);
    const int w = (int)width, h = (int)height;
    const size_t x0w = (size_t)rem_euclid_i(x0, w), y0h = (size_t)rem_euclid_i(y0, h);
    const size_t x1w = (size_t)rem_euclid_i(x0 + 1, w), y1h = (size_t)rem_euclid_i(y0 + 1, h);
    const float fx = px - std::floor(px), fy = py - std::floor(py);
    auto at = [&](size_t yy, size_t xx) {
        const float* p = &texels[(yy * width + xx) * 3];
        return v3(p[0], p[1], p[2]);
    };
    const V3 c0 = lerp(at(y0h, x0w), at(y0h, x1w), fx);
    const V3 c1 = lerp(at(y1h, x0w), at(y1h, x1w), fx);
    return lerp(c0, c1, fy);
}

// ------------------------------------------------------------ TextureSky ----
// sky/texture.rs:67-112
void SkyData::build_cdfs() {
    if (texture.texels.empty()) { width = 1; height = 1; } else { width = texture.width; height = texture.height; }
    conditional_cdfs.assign((size_t)height * (width + 1), 0.0f);
    std::vector<float> row_weights(height, 0.0f);
    for (uint32_t y = 0; y < height; ++y) {
        const float v = ((float)y + 0.5f) / (float)height;
        const float sin_theta = std::sin(PI_F * v);
        const size_t row_offset = (size_t)y * (width + 1);
        conditional_cdfs[row_offset] = 0.0f;
        for (uint32_t x = 0; x < width; ++x) {
            const V3 radiance = texture.sample(((float)x + 0.5f) / (float)width, v);
            const float lum = (0.2126f * radiance.x + 0.7152f * radiance.y) + 0.0722f * radiance.z;
            const float weight = lum * sin_theta;
            conditional_cdfs[row_offset + x + 1] = conditional_cdfs[row_offset + x] + weight;
        }
        row_weights[y] = conditional_cdfs[row_offset + width];
        if (row_weights[y] > 0.0f) {
            const float inv = 1.0f / row_weights[y];
            for (uint32_t x = 0; x <= width; ++x) conditional_cdfs[row_offset + x] *= inv;
        }
    }
    marginal_cdf.assign((size_t)height + 1, 0.0f);
    for (uint32_t y = 0; y < height; ++y) marginal_cdf[y + 1] = marginal_cdf[y] + row_weights[y];
    total_weight = marginal_cdf[height];
    if (total_weight > 0.0f) {
        const float inv = 1.0f / total_weight;
        for (uint32_t y = 0; y <= height; ++y) marginal_cdf[y] *= inv;
    }
}

// ------------------------------------------------------------ blue noise ----
// bluenoise.rs:9-154, 192-210 — void-and-cluster with an xorshift64* stream.
namespace {
struct BnRng {
    uint64_t state;
    explicit BnRng(uint64_t seed) : state(seed + 0x9E3779B97F4A7C15ull) {}
    float next() {
        state ^= state >> 12;
        state ^= state << 25;
        state ^= state >> 27;
        const uint64_t result = state * 0x2545F4914F6CDD1Dull;
        return (float)(result >> 40) / (float)(1ull << 24);
    }
};

void add_energy(std::vector<float>& energy, int size, int px, int py, const std::vector<float>& kernel, int kernel_radius) {
    const float sign = kernel_radius >= 0 ? 1.0f : -1.0f;
    const int kr = kernel_radius < 0 ? -kernel_radius : kernel_radius;
    const int kernel_size = kr * 2 + 1;
    for (int dy = -kr; dy <= kr; ++dy) {
        for (int dx = -kr; dx <= kr; ++dx) {
            const int nx = rem_euclid_i(px + dx, size), ny = rem_euclid_i(py + dy, size);
            const float w = kernel[(size_t)(dy + kr) * kernel_size + (size_t)(dx + kr)];
            energy[(size_t)ny * size + nx] += sign * w;
        }
    }
}

std::vector<float> generate_channel(int size, uint64_t seed) {
    const size_t num_pixels = (size_t)size * size;
    BnRng rng(seed);
    std::vector<float> energy(num_pixels, 0.0f);
    std::vector<char> is_set(num_pixels, 0);
    std::vector<size_t> rank(num_pixels, 0);

    const float sigma = 1.5f;
    const int kernel_radius = (int)std::ceil(sigma * 3.0f);
    const int kernel_size = kernel_radius * 2 + 1;
    std::vector<float> kernel((size_t)kernel_size * kernel_size, 0.0f);
    for (int dy = -kernel_radius; dy <= kernel_radius; ++dy)
        for (int dx = -kernel_radius; dx <= kernel_radius; ++dx) {
            const float d2 = (float)(dx * dx + dy * dy);
            kernel[(size_t)(dy + kernel_radius) * kernel_size + (size_t)(dx + kernel_radius)] =
                std::exp(-d2 / (2.0f * sigma * sigma));
        }

    const size_t initial_count = num_pixels / 10;
    const size_t grid_size = (size_t)std::ceil(std::sqrt((float)initial_count));
    const size_t cell_size = (size_t)size / grid_size;

    for (size_t gy = 0; gy < grid_size; ++gy) {
        for (size_t gx = 0; gx < grid_size; ++gx) {
            if (gy * grid_size + gx >= initial_count) break;
            size_t px = gx * cell_size + (size_t)(rng.next() * (float)cell_size);
            size_t py = gy * cell_size + (size_t)(rng.next() * (float)cell_size);
            px = std::min(px, (size_t)size - 1);
            py = std::min(py, (size_t)size - 1);
            const size_t idx = py * size + px;
            if (!is_set[idx]) {
                is_set[idx] = 1;
                add_energy(energy, size, (int)px, (int)py, kernel, kernel_radius);
            }
        }
    }

    size_t set_count = 0;
    for (char s : is_set) set_count += s ? 1 : 0;
    const size_t target_initial = initial_count / 2;

    auto argmax_set = [&]() {
        float max_energy = std::numeric_limits<float>::lowest();  // f32::MIN
        size_t max_idx = 0;
        for (size_t idx = 0; idx < num_pixels; ++idx)
            if (is_set[idx] && energy[idx] > max_energy) { max_energy = energy[idx]; max_idx = idx; }
        return max_idx;
    };

    while (set_count > target_initial) {
        const size_t max_idx = argmax_set();
        is_set[max_idx] = 0;
        add_energy(energy, size, (int)(max_idx % size), (int)(max_idx / size), kernel, -kernel_radius);
        set_count -= 1;
    }

    size_t current_rank = set_count;
    while (set_count > 0) {
        const size_t max_idx = argmax_set();
        current_rank -= 1;
        rank[max_idx] = current_rank;
        is_set[max_idx] = 0;
        add_energy(energy, size, (int)(max_idx % size), (int)(max_idx / size), kernel, -kernel_radius);
        set_count -= 1;
    }

    current_rank = target_initial;
    while (current_rank < num_pixels) {
        float min_energy = std::numeric_limits<float>::max();
        size_t min_idx = 0;
        for (size_t idx = 0; idx < num_pixels; ++idx)
            if (!is_set[idx] && energy[idx] < min_energy) { min_energy = energy[idx]; min_idx = idx; }
        rank[min_idx] = current_rank;
        current_rank += 1;
        is_set[min_idx] = 1;
        add_energy(energy, size, (int)(min_idx % size), (int)(min_idx / size), kernel, kernel_radius);
    }

    std::vector<float> out(num_pixels);
    for (size_t i = 0; i < num_pixels; ++i) out[i] = (float)rank[i] / (float)(num_pixels - 1);
    return out;
}
}  // namespace

std::vector<float> bluenoise_generate(uint32_t size, uint64_t seed) {
    const std::vector<float> vx = generate_channel((int)size, seed);
    const std::vector<float> vy = generate_channel((int)size, seed + 99999ull);
    std::vector<float> out((size_t)size * size * 2);
    for (size_t i = 0; i < (size_t)size * size; ++i) { out[2 * i] = vx[i]; out[2 * i + 1] = vy[i]; }
    return out;
}

// The render loop's two OnceLock tables (renderloop/parallel.rs:19-28).
static const std::vector<float>& blue_noise_cached(int which) {
    static std::once_flag f0, f1;
    static std::vector<float> t0, t1;
    if (which == 0) { std::call_once(f0, [] { t0 = bluenoise_generate(64, 12345); }); return t0; }
    std::call_once(f1, [] { t1 = bluenoise_generate(64, 67890); });
    return t1;
}

// ------------------------------------------------------------- Instance -----
// accel/mod.rs:84-120
static Aabb compute_world_bounds(const Aabb& lb, const M4& transform) {
    const V3 corners[8] = {
        v3(lb.mn.x, lb.mn.y, lb.mn.z), v3(lb.mx.x, lb.mn.y, lb.mn.z), v3(lb.mn.x, lb.mx.y, lb.mn.z), v3(lb.mx.x, lb.mx.y, lb.mn.z),
        v3(lb.mn.x, lb.mn.y, lb.mx.z), v3(lb.mx.x, lb.mn.y, lb.mx.z), v3(lb.mn.x, lb.mx.y, lb.mx.z), v3(lb.mx.x, lb.mx.y, lb.mx.z)};
    const V3 first = transform_point3(transform, corners[0]);
    V3 mn = first, mx = first;
    for (int i = 1; i < 8; ++i) {
        const V3 w = transform_point3(transform, corners[i]);
        mn = vmin(mn, w);
        mx = vmax(mx, w);
    }
    return Aabb{mn, mx};
}

// ---------------------------------------------------------------- Scene -----
void SceneData::finish() {
    // Tlas::build (accel/mod.rs:166-192)
    instances.clear();
    std::vector<std::pair<uint32_t, Aabb>> inst_bounds;
    for (size_t i = 0; i < instance_data.size(); ++i) {
        const uint32_t blas = instance_data[i].first;
        const M4& transform = instance_data[i].second;
        PtxInstance inst;
        inst.blas_index = blas;
        const M4 inv = inverse(transform);
        const M4 nrm = transpose(inv);
        transform.to_array(inst.transform);
        inv.to_array(inst.inverse_transform);
        nrm.to_array(inst.normal_matrix);
        instances.push_back(inst);
        // Blas::bounds = EMPTY grown by the one item's bounds (accel/mod.rs:38-47)
        Aabb bb = Aabb::empty();
        grow(bb, meshes[blas]->bounds);
        inst_bounds.emplace_back((uint32_t)i, compute_world_bounds(bb, transform));
    }
    tlas = bvh8_build(inst_bounds, 1);

    sky.build_cdfs();
    if (materials.empty()) {  // main.rs:104-107
        PtxMaterial m;
        pth_material_default(&m);
        materials.push_back(m);
    }

    // flatten into the C descriptor
    mesh_descs.resize(meshes.size());
    for (size_t i = 0; i < meshes.size(); ++i) {
        const MeshData& m = *meshes[i];
        PtxMesh& d = mesh_descs[i];
        d.positions = m.positions.data(); d.n_positions = (uint32_t)(m.positions.size() / 3);
        d.normals = m.normals.empty() ? nullptr : m.normals.data(); d.n_normals = (uint32_t)(m.normals.size() / 3);
        d.uvs = m.uvs.empty() ? nullptr : m.uvs.data(); d.n_uvs = (uint32_t)(m.uvs.size() / 2);
        d.indices = m.indices.data(); d.n_triangles = (uint32_t)(m.indices.size() / 3);
        d.material_id = m.material_id;
        d.bvh.nodes = m.bvh.nodes.data(); d.bvh.n_nodes = (uint32_t)m.bvh.nodes.size();
        d.bvh.primitive_indices = m.bvh.primitive_indices.data();
        d.bvh.n_primitive_indices = (uint32_t)m.bvh.primitive_indices.size();
        d.bounds_min[0] = m.bounds.mn.x; d.bounds_min[1] = m.bounds.mn.y; d.bounds_min[2] = m.bounds.mn.z;
        d.bounds_max[0] = m.bounds.mx.x; d.bounds_max[1] = m.bounds.mx.y; d.bounds_max[2] = m.bounds.mx.z;
    }
    tex_descs.resize(textures.size());
    for (size_t i = 0; i < textures.size(); ++i) {
        const TextureData& t = *textures[i];
        tex_descs[i].width = t.width; tex_descs[i].height = t.height;
        tex_descs[i].texels = t.texels.empty() ? nullptr : t.texels.data();
        tex_descs[i].solid[0] = t.solid.x; tex_descs[i].solid[1] = t.solid.y; tex_descs[i].solid[2] = t.solid.z;
    }
    std::memset(&desc, 0, sizeof(desc));
    desc.abi_version = PTX_ABI_VERSION;
    desc.meshes = mesh_descs.data(); desc.n_meshes = (uint32_t)mesh_descs.size();
    desc.instances = instances.data(); desc.n_instances = (uint32_t)instances.size();
    desc.tlas.nodes = tlas.nodes.data(); desc.tlas.n_nodes = (uint32_t)tlas.nodes.size();
    desc.tlas.primitive_indices = tlas.primitive_indices.data();
    desc.tlas.n_primitive_indices = (uint32_t)tlas.primitive_indices.size();
    desc.materials = materials.data(); desc.n_materials = (uint32_t)materials.size();
    desc.textures = tex_descs.data(); desc.n_textures = (uint32_t)tex_descs.size();
    desc.sky.texture.width = sky.texture.width; desc.sky.texture.height = sky.texture.height;
    desc.sky.texture.texels = sky.texture.texels.empty() ? nullptr : sky.texture.texels.data();
    desc.sky.texture.solid[0] = sky.texture.solid.x; desc.sky.texture.solid[1] = sky.texture.solid.y;
    desc.sky.texture.solid[2] = sky.texture.solid.z;
    desc.sky.marginal_cdf = sky.marginal_cdf.data();
    desc.sky.conditional_cdfs = sky.conditional_cdfs.data();
    desc.sky.width = sky.width; desc.sky.height = sky.height;
    desc.sky.total_weight = sky.total_weight; desc.sky.rotation = sky.rotation;
    desc.sun = sun;
    desc.blue_noise_pixel.size = 64; desc.blue_noise_pixel.samples = blue_noise_cached(0).data();
    desc.blue_noise_lens.size = 64; desc.blue_noise_lens.samples = blue_noise_cached(1).data();
}

int32_t SceneData::add_mesh(std::vector<float> positions, std::vector<float> normals, std::vector<float> uvs,
                            std::vector<uint32_t> indices, uint32_t material_id) {
    const size_t npos = positions.size() / 3;
    if (npos == 0 || indices.empty() || indices.size() % 3 != 0) return -1;
    for (uint32_t i : indices) if (i >= npos) return -1;
    if (!normals.empty() && normals.size() / 3 < npos) return -1;
    if (!uvs.empty() && uvs.size() / 2 < npos) return -1;
    auto m = std::make_unique<MeshData>();
    m->positions = std::move(positions); m->normals = std::move(normals); m->uvs = std::move(uvs);
    m->indices = std::move(indices); m->material_id = material_id;
    m->build();
    meshes.push_back(std::move(m));
    return (int32_t)meshes.size() - 1;
}

// light.rs:13-26
void SceneData::set_sun(V3 direction, V3 color) {
    const float angle = 0.25f * (PI_F / 180.0f);  // 0.25_f32.to_radians()
    const float cos_angle = std::cos(angle);
    const float solid_angle = 2.0f * PI_F * (1.0f - cos_angle);
    const V3 d = normalize(direction);
    sun.enabled = 1;
    sun.direction[0] = d.x; sun.direction[1] = d.y; sun.direction[2] = d.z;
    sun.color[0] = color.x; sun.color[1] = color.y; sun.color[2] = color.z;
    sun.cos_angle = cos_angle;
    sun.pdf = 1.0f / solid_angle;
}

// camera.rs:35-60
static void camera_fill(V3 position, float yaw, float pitch, float fov, float aspect, float aperture, float focus,
                        PtxCamera* out) {
    const V3 forward = v3(std::sin(yaw) * std::cos(pitch), std::sin(pitch), -std::cos(yaw) * std::cos(pitch));
    const V3 look_at = position + forward;
    const float fov_rad = fov * (PI_F / 180.0f);
    const M4 vp = mul(perspective_rh(fov_rad, aspect, 0.1f, 1000.0f), look_at_rh(position, look_at, v3(0, 1, 0)));
    inverse(vp).to_array(out->inv_vp_matrix);
    out->aperture = aperture;
    out->focus_distance = focus;
    const V3 right = v3(std::cos(yaw), 0.0f, std::sin(yaw));
    const V3 up = cross(right, forward);
    out->right[0] = right.x; out->right[1] = right.y; out->right[2] = right.z;
    out->up[0] = up.x; out->up[1] = up.y; out->up[2] = up.z;
}

void camera_new(V3 position, V3 look_at, float fov, float aspect, PtxCamera* out) {
    const V3 dir = normalize(look_at - position);
    const float yaw = std::atan2(dir.x, -dir.z);
    const float pitch = std::asin(dir.y);
    camera_fill(position, yaw, pitch, fov, aspect, 0.0f, 10.0f, out);
}

}  // namespace pth

// =============================================================== C ABI =======
using namespace pth;

struct PthBuilder { std::unique_ptr<SceneData> scene; };
struct PthBvh8 { Bvh8Data data; };

extern "C" {

PthBuilder* pth_builder_new(void) {
    auto* b = new PthBuilder();
    b->scene = std::make_unique<SceneData>();
    return b;
}
void pth_builder_free(PthBuilder* b) { delete b; }

int32_t pth_builder_add_mesh(PthBuilder* b, const float* positions, uint32_t n_positions, const float* normals,
                             uint32_t n_normals, const float* uvs, uint32_t n_uvs, const uint32_t* indices,
                             uint32_t n_triangles, uint32_t material_id) {
    if (!b || !b->scene || !positions || !indices) return -1;
    std::vector<float> p(positions, positions + (size_t)n_positions * 3);
    std::vector<float> n, u;
    if (normals && n_normals) n.assign(normals, normals + (size_t)n_normals * 3);
    if (uvs && n_uvs) u.assign(uvs, uvs + (size_t)n_uvs * 2);
    std::vector<uint32_t> idx(indices, indices + (size_t)n_triangles * 3);
    return b->scene->add_mesh(std::move(p), std::move(n), std::move(u), std::move(idx), material_id);
}

int32_t pth_builder_add_instance(PthBuilder* b, uint32_t blas_index, const float transform[16]) {
    if (!b || !b->scene || blas_index >= b->scene->meshes.size()) return -1;
    b->scene->instance_data.emplace_back(blas_index, M4::from_array(transform));
    return (int32_t)b->scene->instance_data.size() - 1;
}

int32_t pth_builder_add_material(PthBuilder* b, const PtxMaterial* m) {
    if (!b || !b->scene || !m) return -1;
    b->scene->materials.push_back(*m);
    return (int32_t)b->scene->materials.size() - 1;
}

int32_t pth_builder_add_texture(PthBuilder* b, uint32_t width, uint32_t height, const float* texels) {
    if (!b || !b->scene || !texels || !width || !height) return -1;
    auto t = std::make_unique<TextureData>();
    t->width = width; t->height = height;
    t->texels.assign(texels, texels + (size_t)width * height * 3);
    b->scene->textures.push_back(std::move(t));
    return (int32_t)b->scene->textures.size() - 1;
}

void pth_builder_set_sky_solid(PthBuilder* b, float r, float g, float bl) {
    b->scene->sky = SkyData();
    b->scene->sky.texture.solid = v3(r, g, bl);
}

void pth_builder_set_sky_image(PthBuilder* b, uint32_t width, uint32_t height, const float* texels, float rotation) {
    b->scene->sky = SkyData();
    b->scene->sky.texture.width = width; b->scene->sky.texture.height = height;
    b->scene->sky.texture.texels.assign(texels, texels + (size_t)width * height * 3);
    b->scene->sky.rotation = rotation;
}

void pth_builder_set_sun(PthBuilder* b, const float direction[3], const float color[3]) {
    b->scene->set_sun(v3(direction[0], direction[1], direction[2]), v3(color[0], color[1], color[2]));
}

PthScene* pth_builder_finish(PthBuilder* b) {
    if (!b || !b->scene) return nullptr;
    auto* s = new PthScene();
    s->scene = std::move(b->scene);
    b->scene = std::make_unique<SceneData>();
    s->scene->finish();
    return s;
}

const PtxSceneDesc* pth_scene_desc(const PthScene* s) { return s ? &s->scene->desc : nullptr; }
void pth_scene_free(PthScene* s) { delete s; }

void pth_scene_stats(const PthScene* s, uint64_t out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!s) return;
    std::vector<char> seen(s->scene->meshes.size(), 0);
    for (const auto& m : s->scene->meshes) { out[0] += m->indices.size() / 3; out[1] += m->bvh.nodes.size(); }
    out[1] += s->scene->tlas.nodes.size();
    out[2] = s->scene->instances.size();
    for (const auto& t : s->scene->textures) out[3] += t->texels.size() * sizeof(float);
    out[3] += s->scene->sky.texture.texels.size() * sizeof(float);
}

void pth_material_default(PtxMaterial* m) {
    // material.rs:38-65
    m->base_color_texture = -1; m->metallic_roughness_texture = -1; m->normal_texture = -1; m->emissive_texture = -1;
    m->base_color_factor[0] = m->base_color_factor[1] = m->base_color_factor[2] = m->base_color_factor[3] = 1.0f;
    m->roughness_factor = 1.0f; m->metallic_factor = 0.0f; m->normal_scale = 1.0f;
    m->emissive_factor[0] = m->emissive_factor[1] = m->emissive_factor[2] = 0.0f;
    m->ior = 1.5f; m->transmission = 0.0f; m->subsurface = 0.0f;
    m->subsurface_radius[0] = 1.0f; m->subsurface_radius[1] = 0.2f; m->subsurface_radius[2] = 0.1f;
    m->subsurface_albedo[0] = 0.98f; m->subsurface_albedo[1] = 0.80f; m->subsurface_albedo[2] = 0.78f;
    m->subsurface_anisotropy = 0.0f;
    m->subsurface_entry_tint[0] = 0.95f; m->subsurface_entry_tint[1] = 0.85f; m->subsurface_entry_tint[2] = 0.80f;
    m->subsurface_exit_tint[0] = 1.0f; m->subsurface_exit_tint[1] = 0.85f; m->subsurface_exit_tint[2] = 0.75f;
}

void pth_camera_new(const float position[3], const float look_at[3], float fov_degrees, float aspect_ratio, PtxCamera* out) {
    camera_new(v3(position[0], position[1], position[2]), v3(look_at[0], look_at[1], look_at[2]), fov_degrees, aspect_ratio, out);
}

void pth_camera_from_angles(const float position[3], float yaw, float pitch, float fov_degrees, float aspect_ratio,
                            float aperture, float focus_distance, PtxCamera* out) {
    camera_fill(v3(position[0], position[1], position[2]), yaw, pitch, fov_degrees, aspect_ratio, aperture, focus_distance, out);
}

void pth_bluenoise_generate(uint32_t size, uint64_t seed, float* out_samples) {
    const std::vector<float> v = bluenoise_generate(size, seed);
    std::memcpy(out_samples, v.data(), v.size() * sizeof(float));
}

PthBvh8* pth_bvh8_build(const float* bounds, uint32_t n, uint32_t max_prims_per_leaf) {
    std::vector<std::pair<uint32_t, Aabb>> pb;
    pb.reserve(n);
    for (uint32_t i = 0; i < n; ++i) {
        const float* b = bounds + (size_t)i * 6;
        pb.emplace_back(i, Aabb{v3(b[0], b[1], b[2]), v3(b[3], b[4], b[5])});
    }
    auto* h = new PthBvh8();
    h->data = bvh8_build(pb, max_prims_per_leaf);
    return h;
}
void pth_bvh8_view(const PthBvh8* h, PtxBvh8* out) {
    out->nodes = h->data.nodes.data(); out->n_nodes = (uint32_t)h->data.nodes.size();
    out->primitive_indices = h->data.primitive_indices.data();
    out->n_primitive_indices = (uint32_t)h->data.primitive_indices.size();
}
void pth_bvh8_free(PthBvh8* h) { delete h; }

}  // extern "C"
