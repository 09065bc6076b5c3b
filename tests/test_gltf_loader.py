"""GLB -> flat scene: the loader's rules (loader/gltf.rs) on a GLB the test writes itself, then the
scene it builds is traced by the oracle."""
import numpy as np
import pytest

from gltf_fixture import write_test_glb


@pytest.fixture(scope="module")
def glb(tmp_path_factory):
    path = tmp_path_factory.mktemp("gltf") / "test.glb"
    return str(path), write_test_glb(str(path))


def test_load_gltf_follows_the_reference_loader(pkg, glb):
    path, want = glb
    g = pkg.gltf.load_gltf(path)
    # one Blas + one instance per TRIANGLES primitive, nodes in document order (node 1 then node 3); LINES skipped
    assert len(g.meshes) == 4 and [b for b, _ in g.instances] == [0, 1, 2, 3]
    quad, tri = g.meshes[0], g.meshes[1]
    assert np.array_equal(quad.positions, want["quad_p"]) and quad.material_id == 1
    assert np.array_equal(quad.indices, want["quad_i"].reshape(-1, 3).astype(np.uint32))
    assert np.array_equal(quad.uvs, want["quad_uv"].astype(np.float32) / np.float32(255.0))      # into_f32
    assert quad.normals.shape == (4, 3)
    assert np.array_equal(tri.positions, want["tri_p"]) and tri.material_id == 0                 # byteStride honoured
    assert np.array_equal(tri.indices, np.array([[0, 1, 2]], dtype=np.uint32))                   # no indices => 0,1,2
    assert tri.normals.shape == (0, 3) and tri.uvs.shape == (0, 2)
    # global transform = root * local; node 3 is outside the scene => identity
    xf = g.instances[0][1]
    p = xf @ np.array([1.0, 0.0, 0.0, 1.0], dtype=np.float32)
    # local: scale 2, rotate +90 deg about Y ((1,0,0) -> (0,0,-1)), translate (0,0,-1); then root translate (1,2,3)
    assert np.allclose(p[:3], [1.0, 2.0, 3.0 - 1.0 - 2.0], atol=1e-6)
    assert np.array_equal(g.instances[0][1], g.instances[1][1])
    assert np.array_equal(g.instances[2][1], np.eye(4, dtype=np.float32))
    # textures: one per IMAGE; image 1 (base colour) is sRGB -> linear, image 0 (normal map) stays linear, grey -> 0.5
    assert len(g.textures) == 3
    assert g.textures[0].shape == (4, 8, 3) and g.textures[1].shape == (8, 4, 3) and g.textures[2].shape == (2, 2, 3)
    assert np.array_equal(g.textures[0], want["img_lin"].astype(np.float32) / np.float32(255.0))
    c = want["img_srgb"][..., :3].astype(np.float32) / np.float32(255.0)
    lin = np.where(c <= 0.04045, c / 12.92, ((c.astype(np.float64) + 0.055) / 1.055) ** 2.4)
    assert np.allclose(g.textures[1], lin, rtol=2e-6, atol=1e-9)
    assert (g.textures[2] == 0.5).all() and any("0.5 grey" in n for n in g.notes)
    # materials carry IMAGE indices through the texture -> image map
    m0, m1 = g.materials
    assert m0.base_color_texture == -1 and list(m0.base_color_factor) == pytest.approx([0.8, 0.7, 0.6, 1.0])
    assert m0.metallic_factor == 0.0 and m0.roughness_factor == 0.5 and m0.ior == 1.5 and m0.transmission == 0.0
    assert (m1.base_color_texture, m1.normal_texture, m1.metallic_roughness_texture, m1.emissive_texture) == (1, 0, 2, -1)
    assert m1.normal_scale == 0.75 and m1.transmission == 0.25 and m1.ior == pytest.approx(1.33)
    assert list(m1.emissive_factor) == pytest.approx([0.1, 0.2, 0.3]) and m1.metallic_factor == 1.0 and m1.roughness_factor == 1.0
    # camera: position = column 3, forward = -column 2, fov in degrees
    assert np.array_equal(g.camera["position"], np.array([0, 1.5, 6], dtype=np.float32))
    assert np.allclose(g.camera["look_at"], [0, 1.5, 5]) and g.camera["fov_degrees"] == pytest.approx(np.degrees(0.8), rel=1e-6)


def test_gltf_scene_traces(pkg, oracle_det, glb):
    path, _ = glb
    g = pkg.gltf.load_gltf(path)
    sc = pkg.gltf.build_scene(g)
    assert sc.stats()["instances"] == 4 and sc.stats()["triangles"] == 2 * (2 + 1)
    cam = pkg.gltf.camera_for(g, 64 / 36)
    osc = oracle_det.scene(sc)
    rays = osc.primary_rays(cam, 64, 36, 0)
    hits = osc.trace_closest(rays)
    hit = hits["inst_idx"] != pkg._abi.PTX_MISS
    assert 0.01 < hit.mean() < 0.9 and set(np.unique(hits["inst_idx"][hit])) <= {0, 1, 2, 3}
    img = osc.render_pass(cam, 64, 36, 0, 8)
    assert np.isfinite(img).all() and img.mean() > 0.05


def test_gltf_errors(pkg, tmp_path):
    p = tmp_path / "bad.glb"
    p.write_bytes(b"glTF" + (1).to_bytes(4, "little") + (12).to_bytes(4, "little"))
    with pytest.raises(pkg.gltf.GltfError):
        pkg.gltf.load_gltf(str(p))


def test_matrix_only_nodes_pass_through_bit_for_bit(pkg, tmp_path):
    """Nodes that carry a `matrix` (no TRS decomposition involved): the global transform is `parent * local`
    (loader/gltf.rs:210-235) in glam's f32 operation order — ((a.x_axis*b.x + a.y_axis*b.y) + a.z_axis*b.z) + a.w_axis*b.w
    per column — and must come out bit for bit."""
    import base64
    import json
    rng = np.random.default_rng(4)
    a = np.eye(4, dtype=np.float32); a[:3, :] = rng.normal(size=(3, 4)).astype(np.float32)      # row-major, affine
    b = np.eye(4, dtype=np.float32); b[:3, :] = rng.normal(size=(3, 4)).astype(np.float32)
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    buf = pos.tobytes()
    doc = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
        "nodes": [{"matrix": a.T.reshape(-1).tolist(), "children": [1]}, {"matrix": b.T.reshape(-1).tolist(), "mesh": 0}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0}}]}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3", "min": [0, 0, 0], "max": [1, 1, 0]}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": len(buf)}],
        "buffers": [{"byteLength": len(buf), "uri": "data:application/octet-stream;base64," + base64.b64encode(buf).decode()}],
    }
    path = tmp_path / "m.gltf"
    path.write_text(json.dumps(doc))
    g = pkg.gltf.load_gltf(str(path))
    # JSON round-trips f32 values exactly when they are written as the shortest repr of the double they convert to
    ac, bc = a.T.copy(), b.T.copy()                    # columns as rows
    want = np.zeros((4, 4), dtype=np.float32)          # want[j] = column j of a * b
    for j in range(4):
        col = ac[0] * bc[j][0]
        col = col + ac[1] * bc[j][1]
        col = col + ac[2] * bc[j][2]
        col = col + ac[3] * bc[j][3]
        want[j] = col
    got = np.asarray(g.instances[0][1], dtype=np.float32)      # row-major
    assert np.array_equal(got.T.view(np.uint32), want.view(np.uint32))
