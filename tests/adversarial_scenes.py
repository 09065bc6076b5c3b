"""Scenes and ray batches built to attack the presumption the nearest-first walk and the forwarded bound rest on
(csrc/device/pt_traverse.h "FAST mode"): that a computed hit distance does not lie far in front of the entry distance
of the boxes around the triangle.  Moeller-Trumbore (shape/triangle_mesh.rs:63-100) loses all accuracy when
`a = e1 . (d x e2)` is a difference of nearly equal products: needle / sliver triangles (e1 almost parallel to e2) and
rays that graze a triangle's plane.  Shared by tests/test_gpu_adversarial.py and tools/adversarial_probe.py.
"""
import numpy as np


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _perp(u, rng):
    r = rng.normal(size=u.shape)
    r -= (r * u).sum(-1, keepdims=True) * u
    return _unit(r)


def rot(axis, angle):
    a = np.asarray(axis, dtype=np.float64)
    a /= np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def affine(lin, t):
    m = np.eye(4)
    m[:3, :3], m[:3, 3] = lin, t
    return m


def quad(p, ex, ez, nx=1, nz=1):
    """A parallelogram p + s ex + t ez tessellated nx x nz (two triangles per cell)."""
    p, ex, ez = (np.asarray(a, dtype=np.float64) for a in (p, ex, ez))
    pos = [p + ex * (i / nx) + ez * (j / nz) for i in range(nx + 1) for j in range(nz + 1)]
    idx = []
    for i in range(nx):
        for j in range(nz):
            a, b, c, d = i * (nz + 1) + j, (i + 1) * (nz + 1) + j, (i + 1) * (nz + 1) + j + 1, i * (nz + 1) + j + 1
            idx += [(a, b, c), (a, c, d)]
    return np.array(pos), np.array(idx, dtype=np.int64)


def merge(parts):
    pos, idx, base = [], [], 0
    for p, i in parts:
        pos.append(p)
        idx.append(i + base)
        base += len(p)
    return np.concatenate(pos).astype(np.float32), np.concatenate(idx).astype(np.uint32)


def needles(rng, n, aspects=(1e4, 1e5, 1e6), extent=4.0, lengths=(0.1, 3.0), degenerate=0.05):
    """n needle triangles: length L, width L / aspect, random pose inside a cube; the vertex order is random, so the
    angle at v0 (the one Moeller-Trumbore's `a` depends on) is the sharp tip for a third of them.  A share
    `degenerate` are so small that |a| sits just above the reference's 1e-8 cut for rays that meet them head on."""
    c = rng.uniform(-extent, extent, size=(n, 3))
    u = _unit(rng.normal(size=(n, 3)))
    v = _perp(u, rng)
    L = np.exp(rng.uniform(np.log(lengths[0]), np.log(lengths[1]), size=(n, 1)))
    asp = rng.choice(aspects, size=(n, 1))
    small = rng.uniform(size=(n, 1)) < degenerate
    L = np.where(small, 3e-4, L)                 # |e1||e2| ~ 1e-7 / aspect ... : |a| ~ 1e-8 .. 1e-7 at normal incidence
    asp = np.where(small, rng.choice([1.0, 3.0, 10.0], size=(n, 1)), asp)
    p0, p1, p2 = c - u * L / 2, c + u * L / 2, c + u * L / 2 + v * (L / asp)
    tri = np.stack([p0, p1, p2], axis=1)
    order = np.array([rng.permutation(3) for _ in range(n)])
    tri = np.take_along_axis(tri, order[:, :, None], axis=1)
    pos = tri.reshape(-1, 3)
    idx = np.arange(3 * n, dtype=np.int64).reshape(n, 3)
    return pos, idx, (p0, p1, p2)


def grazing_rays(abi, rng, tris, per_tri=4, angles=(1e-3, 1e-4, 1e-5, 1e-6), dists=(0.3, 3.0, 30.0), finite=0.3):
    """Rays aimed at interior points of the given triangles under grazing angles (radians between the ray and the
    triangle's plane).  tris = (p0, p1, p2) arrays.  A share `finite` are segments that end within +-3 % of the
    target."""
    p0, p1, p2 = tris
    n = len(p0)
    k = n * per_tri
    ti = np.repeat(np.arange(n), per_tri)
    b = rng.dirichlet((1.0, 1.0, 1.0), size=k)
    target = p0[ti] * b[:, :1] + p1[ti] * b[:, 1:2] + p2[ti] * b[:, 2:3]
    nrm = _unit(np.cross(p1 - p0, p2 - p0))[ti]
    inplane = _perp(nrm, rng)
    # half of them along the needle's own axis (the direction in which the thin triangle is long)
    axis = _unit(p1 - p0)[ti]
    along = rng.uniform(size=(k, 1)) < 0.5
    inplane = np.where(along, _unit(axis + 0.02 * inplane), inplane)
    inplane = _unit(inplane - (inplane * nrm).sum(-1, keepdims=True) * nrm)
    th = rng.choice(angles, size=(k, 1)) * rng.choice([-1.0, 1.0], size=(k, 1))
    d = _unit(np.cos(th) * inplane + np.sin(th) * nrm)
    dist = rng.choice(dists, size=(k, 1)) * rng.uniform(0.5, 1.5, size=(k, 1))
    o = target - d * dist
    rays = np.zeros(k, dtype=abi.ray_dtype())
    rays["origin"], rays["direction"] = o.astype(np.float32), d.astype(np.float32)
    tm = np.where(rng.uniform(size=k) < finite, dist[:, 0] * rng.uniform(0.97, 1.03, size=k), np.inf)
    rays["t_max"] = tm.astype(np.float32)
    return rays


def random_rays(abi, rng, n, extent=5.0):
    o = rng.uniform(-extent, extent, size=(n, 3))
    t = rng.uniform(-extent * 0.8, extent * 0.8, size=(n, 3))
    d = _unit(t - o)
    rays = np.zeros(n, dtype=abi.ray_dtype())
    rays["origin"], rays["direction"] = o.astype(np.float32), d.astype(np.float32)
    rays["t_max"] = np.inf
    return rays


def needle_scene(pkg, rng, n_needles=6000, instanced=True, far=False, aspects=(1e4, 1e5, 1e6), degenerate=0.05):
    """A cloud of needles and near-degenerate triangles, with a few well-shaped walls among them (so that many rays
    have a well-conditioned nearer hit while ill-conditioned triangles wait behind it), as one mesh; optionally
    instanced under rotations / anisotropic scales / far translations."""
    pos_n, idx_n, tris = needles(rng, n_needles, aspects=aspects, degenerate=degenerate)
    walls = [quad((-4, -4, z), (8, 0, 0), (0, 8, 0), 6, 6) for z in (-2.5, 0.0, 2.0)]
    walls += [quad((x, -4, -4), (0, 8, 0), (0, 0, 8), 5, 5) for x in (-3.0, 1.0)]
    pos, idx = merge([(pos_n, idx_n)] + walls)
    b = pkg.SceneBuilder()
    b.add_material(pkg.default_material())
    mesh = b.add_mesh(pos, idx)
    transforms = [np.eye(4)]
    if instanced:
        transforms += [affine(rot((1, 2, 3), 0.7), (0.3, -0.2, 0.1)), affine(rot((0, 1, 0), 2.1) @ np.diag([1.5, 0.7, 1.0]), (9.0, 0.5, -1.0))]
    if far:
        transforms = [affine(m[:3, :3], m[:3, 3] + np.array([3000.0, -2000.0, 1500.0])) for m in transforms]
    for m in transforms:
        b.add_instance(mesh, np.asarray(m, dtype=np.float32))
    b.set_sky_solid(0.5, 0.6, 0.8)
    return b.build(), tris, transforms


def world_tris(tris, m):
    return tuple(p @ m[:3, :3].T + m[:3, 3] for p in tris)


def sliver_layers_scene(pkg, rng, aspect=1e4, layers=(0.0, 0.0, 1e-6, 3e-5, 1e-3), strips=120, tilt=True):
    """Near-coincident layers of long thin strips (each strip two sliver triangles of the given aspect), triangles
    shuffled; tilted so that no plane is axis-aligned."""
    L = 4.0
    w = L / aspect
    parts, tris = [], []
    for y in layers:
        for s in range(strips):
            p, i = quad((-L / 2, y, s * w - strips * w / 2), (L, 0, 0), (0, 0, w), 1, 1)
            parts.append((p, i))
    pos, idx = merge(parts)
    idx = idx[rng.permutation(len(idx))]
    lin = rot((1, 0.3, 0.2), 0.6) if tilt else np.eye(3)
    b = pkg.SceneBuilder()
    b.add_material(pkg.default_material())
    mesh = b.add_mesh(pos, idx)
    transforms = [affine(lin, (0.0, 0.0, 0.0)), affine(lin, lin @ np.array([0.0, 2e-6, 0.0])), affine(lin @ rot((0, 1, 0), 0.01), (0.0, 0.3, 0.0))]
    for m in transforms:
        b.add_instance(mesh, np.asarray(m, dtype=np.float32))
    b.set_sky_solid(0.5, 0.6, 0.8)
    p64 = pos.astype(np.float64)
    tri = (p64[idx[:, 0]], p64[idx[:, 1]], p64[idx[:, 2]])
    return b.build(), tri, transforms


def floor_scene(pkg, rng, n=48, tilt=False, offset=(0.0, 0.0, 0.0)):
    """A finely tessellated floor of well-shaped triangles with boxes standing on it: rays that start on the floor and
    skim along it are grazing rays for every floor triangle ahead of them, also the ones behind the box they hit."""
    parts = [quad((-10, 0, -10), (20, 0, 0), (0, 0, 20), n, n)]
    for (x, z, sx, sz, h) in [(-3, -2, 2, 1, 2), (2, 3, 1, 3, 1.5), (4, -5, 3, 0.5, 3), (-6, 4, 0.5, 0.5, 4), (0, 0, 1, 1, 1)]:
        parts += [quad((x, 0, z), (sx, 0, 0), (0, h, 0), 2, 2), quad((x, 0, z + sz), (sx, 0, 0), (0, h, 0), 2, 2),
                  quad((x, 0, z), (0, 0, sz), (0, h, 0), 2, 2), quad((x + sx, 0, z), (0, 0, sz), (0, h, 0), 2, 2),
                  quad((x, h, z), (sx, 0, 0), (0, 0, sz), 2, 2)]
    pos, idx = merge(parts)
    lin = rot((1, 0.2, 0.5), 0.5) if tilt else np.eye(3)
    m = affine(lin, offset)
    b = pkg.SceneBuilder()
    b.add_material(pkg.default_material())
    mesh = b.add_mesh(pos, idx)
    b.add_instance(mesh, m.astype(np.float32))
    b.set_sky_solid(0.5, 0.6, 0.8)
    return b.build(), m


def skimming_rays(abi, rng, m, k, angles=(1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 0.0)):
    """Rays that start 0.001 above the floor (where a bounce ray starts: hit_point + geo_n * 0.001,
    pathtracer.rs:316) and leave under a grazing angle, up or down."""
    p = np.stack([rng.uniform(-9.5, 9.5, k), np.full(k, 0.001), rng.uniform(-9.5, 9.5, k)], axis=1)
    az = rng.uniform(0, 2 * np.pi, k)
    th = rng.choice(angles, size=k) * rng.choice([-1.0, 1.0], size=k)
    d = np.stack([np.cos(th) * np.cos(az), np.sin(th), np.cos(th) * np.sin(az)], axis=1)
    o = p @ m[:3, :3].T + m[:3, 3]
    dw = d @ m[:3, :3].T
    rays = np.zeros(k, dtype=abi.ray_dtype())
    rays["origin"], rays["direction"] = o.astype(np.float32), dw.astype(np.float32)
    rays["t_max"] = np.inf
    return rays


def diff_rows(got, want):
    n = len(want)
    return np.flatnonzero((got.view(np.uint8).reshape(n, -1) != want.view(np.uint8).reshape(n, -1)).any(axis=1))
