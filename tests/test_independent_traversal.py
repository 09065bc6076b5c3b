"""Second, independent restatement of the traversal (SURVEY.md §8(c)(i),(ii)) — written in numpy-float32 Python
straight from the reference text (accel/bvh.rs:171-223, 308-415; accel/mod.rs:122-146, 194-210;
shape/triangle_mesh.rs:63-100), not from the C++ oracle — plus a brute-force all-triangles check without any BVH.
Both must agree with the C++ oracle, which guards against one shared misreading."""
import ctypes as C

import numpy as np
import pytest

f32 = np.float32
INF = f32(np.inf)
LEAF = 1 << 31
np.seterr(all="ignore")


def mn(a, b):   # _mm256_min_ps: second operand if either is NaN
    return a if a < b else b


def mx(a, b):
    return a if a > b else b


class Bvh:
    def __init__(self, view):
        n = view.n_nodes
        raw = np.ctypeslib.as_array(C.cast(view.nodes, C.POINTER(C.c_uint32)), shape=(n, 64)).copy() if n else np.zeros((0, 64), np.uint32)
        self.f = raw.view(np.float32)
        self.u = raw
        self.prims = np.ctypeslib.as_array(view.primitive_indices, shape=(view.n_primitive_indices,)).copy() if view.n_primitive_indices else np.zeros(0, np.uint32)

    def intersect(self, i, o, inv, tmin, tmax):
        mask, dist = 0, [f32(0)] * 8
        for s in range(8):
            near, far = [], []
            for ax in range(3):
                lo, hi = self.f[i, 16 * ax + s], self.f[i, 16 * ax + 8 + s]
                t1 = f32(f32(lo - o[ax]) * inv[ax])
                t2 = f32(f32(hi - o[ax]) * inv[ax])
                near.append(mn(t1, t2)); far.append(mx(t1, t2))
            te = mx(tmin, mx(near[0], mx(near[1], near[2])))
            tx = mn(tmax, mn(far[0], mn(far[1], far[2])))
            dist[s] = te
            if te <= tx:
                mask |= 1 << s
        return mask, dist

    def closest(self, o, d, tmax, test):
        if len(self.u) == 0:
            return None
        inv = [f32(1) / d[0], f32(1) / d[1], f32(1) / d[2]]
        best, ct, stack = None, tmax, [(0, f32(0))]
        while stack:
            idx, dist = stack.pop()
            if dist >= ct:
                continue
            mask, ds = self.intersect(idx, o, inv, f32(0.001), ct)
            mask &= int(self.u[idx, 56] & 0xFF)
            for s in range(8):
                if not (mask >> s) & 1 or ds[s] >= ct:
                    continue
                w = int(self.u[idx, 48 + s])
                if w & LEAF:
                    start, cnt = w & 0xFFFFFF, ((w >> 24) & 0x7F) + 1
                    for p in self.prims[start:start + cnt]:
                        r = test(int(p))
                        if r is not None and r[0] < ct:
                            ct, best = r[0], r
                else:
                    stack.append((w, ds[s]))
        return best

    def any(self, o, d, tmax, test):
        if len(self.u) == 0:
            return False
        inv = [f32(1) / d[0], f32(1) / d[1], f32(1) / d[2]]
        stack = [0]
        while stack:
            idx = stack.pop()
            mask, _ = self.intersect(idx, o, inv, f32(0.001), tmax)
            mask &= int(self.u[idx, 56] & 0xFF)
            for s in range(8):
                if not (mask >> s) & 1:
                    continue
                w = int(self.u[idx, 48 + s])
                if w & LEAF:
                    start, cnt = w & 0xFFFFFF, ((w >> 24) & 0x7F) + 1
                    for p in self.prims[start:start + cnt]:
                        if test(int(p)):
                            return True
                else:
                    stack.append(w)
        return False


def dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def cross(a, b):
    return [f32(f32(a[1] * b[2]) - f32(b[1] * a[2])), f32(f32(a[2] * b[0]) - f32(b[2] * a[0])), f32(f32(a[0] * b[1]) - f32(b[0] * a[1]))]


def sub(a, b):
    return [f32(a[0] - b[0]), f32(a[1] - b[1]), f32(a[2] - b[2])]


def xf(m, p, w):
    r = []
    for k in range(3):
        v = f32(f32(f32(m[k] * p[0]) + f32(m[4 + k] * p[1])) + f32(m[8 + k] * p[2]))
        r.append(f32(v + m[12 + k]) if w else v)
    return r


class PyScene:
    def __init__(self, desc):
        d = desc.contents
        self.tlas = Bvh(d.tlas)
        self.inst, self.mesh = [], []
        for i in range(d.n_instances):
            it = d.instances[i]
            self.inst.append((it.blas_index, np.array(list(it.transform), f32), np.array(list(it.inverse_transform), f32)))
        for i in range(d.n_meshes):
            m = d.meshes[i]
            pos = np.ctypeslib.as_array(m.positions, shape=(m.n_positions, 3)).copy()
            idx = np.ctypeslib.as_array(m.indices, shape=(m.n_triangles, 3)).copy()
            self.mesh.append((Bvh(m.bvh), pos, idx, np.array(list(m.bounds_min), f32), np.array(list(m.bounds_max), f32)))

    @staticmethod
    def tri(pos, idx, t, o, d):
        v0, v1, v2 = pos[idx[t, 0]], pos[idx[t, 1]], pos[idx[t, 2]]
        e1, e2 = sub(v1, v0), sub(v2, v0)
        h = cross(d, e2)
        a = dot(e1, h)
        if abs(a) < f32(1e-8):
            return None
        f = f32(1) / a
        s = sub(o, v0)
        u = f32(f * dot(s, h))
        if not (f32(0) <= u <= f32(1)):
            return None
        q = cross(s, e1)
        v = f32(f * dot(d, q))
        if v < 0 or f32(u + v) > f32(1):
            return None
        tt = f32(f * dot(e2, q))
        if tt >= f32(0.001) and tt <= INF:
            return (tt, t, u, v)
        return None

    def blas_box(self, m, lo, ld):
        inv = [f32(1) / ld[0], f32(1) / ld[1], f32(1) / ld[2]]
        near, far = [], []
        for ax in range(3):
            t1 = f32(f32(m[3][ax] - lo[ax]) * inv[ax]); t2 = f32(f32(m[4][ax] - lo[ax]) * inv[ax])
            near.append(mn(t1, t2)); far.append(mx(t1, t2))
        te = mx(f32(0.001), mx(near[0], mx(near[1], near[2])))
        tx = mn(INF, mn(far[0], mn(far[1], far[2])))
        return te <= tx, te

    def hit(self, o, d, tmax):
        def inst_test(i):
            blas, M, Mi = self.inst[i]
            m = self.mesh[blas]
            lo, ld = xf(Mi, o, True), xf(Mi, d, False)
            ok, te = self.blas_box(m, lo, ld)
            if not ok or te >= INF:
                return None
            r = m[0].closest(lo, ld, INF, lambda t: self.tri(m[1], m[2], t, lo, ld))
            if r is None:
                return None
            lp = [f32(lo[k] + f32(r[0] * ld[k])) for k in range(3)]
            wp = xf(M, lp, True)
            tw = f32(dot(sub(wp, o), d) / dot(d, d))
            return (tw, i, r[1], r[2], r[3])
        return self.tlas.closest(o, d, tmax, inst_test)

    def hit_any(self, o, d, tmax):
        def inst_test(i):
            blas, M, Mi = self.inst[i]
            m = self.mesh[blas]
            lo, ld = xf(Mi, o, True), xf(Mi, d, False)
            ok, _ = self.blas_box(m, lo, ld)
            return ok and m[0].any(lo, ld, INF, lambda t: self.tri(m[1], m[2], t, lo, ld) is not None)
        return self.tlas.any(o, d, tmax, inst_test)

    def brute(self, o, d, tmax):
        """No BVH at all: every triangle of every instance, nearest world t."""
        best = None
        for i, (blas, M, Mi) in enumerate(self.inst):
            m = self.mesh[blas]
            lo, ld = xf(Mi, o, True), xf(Mi, d, False)
            loc = None
            for t in range(len(m[2])):
                r = self.tri(m[1], m[2], t, lo, ld)
                if r is not None and (loc is None or r[0] < loc[0]):
                    loc = r
            if loc is None:
                continue
            lp = [f32(lo[k] + f32(loc[0] * ld[k])) for k in range(3)]
            tw = f32(dot(sub(xf(M, lp, True), o), d) / dot(d, d))
            if tw < tmax and (best is None or tw < best[0]):
                best = (tw, i, loc[1])
        return best


@pytest.mark.parametrize("name,flags,detail,n_rays", [("cornell", 0, 1.0, 160), ("instanced_stress", 0, 0.004, 100), ("glass_test", 0, 0.02, 60)])
def test_numpy_restatement_and_brute_force_agree_with_oracle(pkg, oracle_det, name, flags, detail, n_rays):
    abi = pkg._abi
    sc = pkg.Scene.procedural(name, 1, detail, flags)
    osc = oracle_det.scene(sc)
    py = PyScene(sc.desc)
    w, h = 48, 27
    cam = sc.default_camera(w / h)
    rays = osc.primary_rays(cam, w, h, 0)
    rng = np.random.default_rng(1)
    pick = rng.choice(len(rays), n_rays, replace=False)
    rays = rays[pick].copy()
    # second half: bounce rays from the hit points with a finite t_max on some
    hc = osc.trace_closest(rays)
    k = n_rays // 2
    ok = hc["inst_idx"][:k] != abi.PTX_MISS
    dirs = rng.normal(size=(k, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    rays["origin"][:k][ok] = (rays["origin"][:k] + rays["direction"][:k] * hc["t"][:k, None] + dirs * f32(1e-3))[ok]
    rays["direction"][:k][ok] = dirs[ok]
    rays["t_max"][:k:3] = f32(1.5)
    hc, ac = osc.trace_closest(rays), osc.trace_any(rays)
    for i, r in enumerate(rays):
        o, d, tm = list(r["origin"]), list(r["direction"]), f32(r["t_max"])
        p = py.hit(o, d, tm)
        if p is None:
            assert hc["inst_idx"][i] == abi.PTX_MISS
        else:
            assert (hc["inst_idx"][i], hc["tri_idx"][i]) == (p[1], p[2])
            assert hc["t"][i].view(np.uint32) == f32(p[0]).view(np.uint32) and hc["u"][i] == p[3] and hc["v"][i] == p[4]
        assert bool(ac[i]) == py.hit_any(o, d, tm)
        b = py.brute(o, d, tm)
        if b is None:
            assert hc["inst_idx"][i] == abi.PTX_MISS
        else:
            assert hc["inst_idx"][i] != abi.PTX_MISS
            # ties / near-ties may resolve to a neighbouring triangle; the distance must agree
            assert abs(float(hc["t"][i]) - float(b[0])) <= 1e-4 * max(1.0, float(b[0]))
            if abs(float(hc["t"][i]) - float(b[0])) == 0.0 and (hc["inst_idx"][i], hc["tri_idx"][i]) != (b[1], b[2]):
                pass   # exact tie between two triangles: the BVH's first-encountered rule decides
