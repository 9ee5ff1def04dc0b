"""GPU parity tests of the closest-hit trees that keep the reference's visiting order (occlusion_bvh.cpp rs_build_ordered_bvh,
rs_scene.h walk_ordered_tree): the wave-level service the multi-bounce kernels use for their bounce rays must return what
DevScene::intersect (src/scene.h:245-284) returns -- primitive, material, position and normal bit for bit -- for every ray:
against the library's literal per-lane walk of the reference's tree, against the same service with the trees switched off, and
against the oracle."""
import numpy as np
import pytest

from tests.common import bits_equal, get_scene, hip_scene, oracle_scene

pytestmark = pytest.mark.gpu


def closest_like_rays(sd, n, seed):
    """Rays of the kinds the bounce passes produce and the corner cases of an order-dependent closest hit: random rays inside
    the scene, bounce rays (origin on a surface, offset 1e-5 along a direction of the hemisphere), rays aimed at triangle
    vertices and at points of shared edges (equal hit distances on several triangles: the first in the reference's order wins),
    rays inside a triangle's plane (hit distances that are pure rounding), axis-aligned / near-zero components (reference walk),
    origins far outside the scene (beyond the grid's reach), NaN directions."""
    rng = np.random.default_rng(seed)
    v = sd.vertices.reshape(-1, 3, 3).astype(np.float64)
    lo, hi = v.reshape(-1, 3).min(0), v.reshape(-1, 3).max(0)

    def points_on(k):
        t = v[rng.integers(0, len(v), k)]
        u = rng.uniform(size=(k, 2)); flip = u.sum(1) > 1; u[flip] = 1 - u[flip]
        p = t[:, 0] * (1 - u.sum(1))[:, None] + t[:, 1] * u[:, :1] + t[:, 2] * u[:, 1:]
        nrm = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
        return p, nrm, t

    def unit(d):
        return d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)

    o = rng.uniform(lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo), (n, 3))
    d = unit(rng.normal(size=(n, 3)))
    k = n // 8
    # bounce rays
    p, nrm, _ = points_on(2 * k)
    dd = unit(rng.normal(size=(2 * k, 3))); dd *= np.sign(np.einsum("ij,ij->i", dd, nrm))[:, None]
    o[k:3 * k] = p + dd * 1e-5; d[k:3 * k] = dd
    # at vertices and at points of edges
    _, _, t = points_on(k)
    w = rng.uniform(size=(k, 1)); w[: k // 2] = 0.0
    target = t[:, 0] * (1 - w) + t[:, 1] * w
    d[3 * k:4 * k] = unit(target - o[3 * k:4 * k])
    # inside a triangle's plane, starting outside the triangle
    g, _, tg = points_on(k)
    o[4 * k:5 * k] = g + (tg[:, 0] - g) * 3.0; d[4 * k:5 * k] = unit((g + (tg[:, 1] - g) * 3.0) - o[4 * k:5 * k])
    # special directions
    s = k // 3
    ax = rng.integers(0, 3, s)
    d[5 * k:5 * k + s] = 0; d[np.arange(5 * k, 5 * k + s), ax] = rng.choice([-1.0, 1.0], s)
    d[5 * k + s:5 * k + 2 * s, 0] = rng.uniform(-1e-6, 1e-6, s)
    d[5 * k + 2 * s:5 * k + 3 * s, 1] = 0.0
    d[5 * k:6 * k] = unit(d[5 * k:6 * k])
    # far origins, aimed at the scene
    o[6 * k:6 * k + 64] = hi + (hi - lo) * rng.uniform(6, 40, (64, 1))
    d[6 * k:6 * k + 64] = unit(rng.uniform(lo, hi, (64, 3)) - o[6 * k:6 * k + 64])
    rays = np.ascontiguousarray(np.concatenate([o, d], 1), np.float32)
    rays[-8:, 3:] = np.nan
    return rays


def _same(a, b):
    pa, ma, xa, na = [t.cpu().numpy() for t in a]
    pb, mb, xb, nb = [t.cpu().numpy() for t in b]
    assert np.array_equal(pa, pb), int((pa != pb).sum())
    hit = pa >= 0
    assert np.array_equal(ma[hit], mb[hit])
    assert bits_equal(xa[hit], xb[hit]) and bits_equal(na[hit], nb[hit])
    return hit


@pytest.mark.parametrize("name,n", [("cornell", 200000), ("sponza:0.1", 400000), ("bistro:0.05", 400000)])
def test_ordered_tree_equals_reference_walk(hip, name, n):
    import torch
    sd = get_scene(name)
    hsc = hip_scene(hip, sd)
    rays = closest_like_rays(sd, n, 31)
    dr = torch.from_numpy(rays).cuda()
    ref = hip.trace_closest(hsc, dr)                       # DevScene::intersect literally, lane by lane
    assert hip.set_ordered_tree(hsc, True), "the closest-hit trees were not built for a scene from rs_build_bvh"
    fast = hip.trace_closest_wave(hsc, dr)
    hit = _same(ref, fast)
    assert 0.3 < hit.mean() <= 1.0
    assert hip.set_ordered_tree(hsc, False)
    _same(ref, hip.trace_closest_wave(hsc, dr))            # the same service on the reference's tree
    hip.set_ordered_tree(hsc, True)
    sub = rays[::20]
    prim, mat, pos, nrm, _ = oracle_scene(sd).intersect(sub)
    gp, gm, gpos, gn = [t.cpu().numpy()[::20] for t in fast]
    assert np.array_equal(prim, gp)
    h = prim >= 0
    assert np.array_equal(mat[h], gm[h]) and bits_equal(pos[h], gpos[h]) and bits_equal(nrm[h], gn[h])


@pytest.mark.parametrize("name", ["sponza:1.0", "bistro:1.0"])
def test_ordered_tree_on_the_full_scenes(hip, name):
    """200 000 rays of every kind on the FULL scenes (262 144 / 2.83 M triangles): the closest-hit trees = the library's
    reference walk for every ray, = the oracle on every 10th."""
    import torch
    sd = get_scene(name)
    hsc = hip_scene(hip, sd)
    assert hip.set_ordered_tree(hsc, True)
    rays = closest_like_rays(sd, 200000, 32)
    dr = torch.from_numpy(rays).cuda()
    fast = hip.trace_closest_wave(hsc, dr)
    hit = _same(hip.trace_closest(hsc, dr), fast)
    assert hit.mean() > 0.3
    sub = rays[::10]
    prim, mat, pos, nrm, _ = oracle_scene(sd).intersect(sub)
    gp, gm, gpos, gn = [t.cpu().numpy()[::10] for t in fast]
    assert np.array_equal(prim, gp)
    h = prim >= 0
    assert np.array_equal(mat[h], gm[h]) and bits_equal(pos[h], gpos[h]) and bits_equal(nrm[h], gn[h])


def test_ordered_tree_is_off_for_tables_without_mirrored_orders(hip):
    """rs_scene_create accepts any table of threaded orders.  The closest-hit trees need the odd orders to be the mirror images of
    the even ones (src/bvh.cpp:186-190); a table where they are not keeps the reference walk, and the results are the reference's."""
    import torch
    sd = get_scene("sponza:0.03")
    base = hip_scene(hip, sd)
    t = base.host_desc()
    t["nodes"] = t["nodes"].copy()
    t["nodes"][1] = t["nodes"][0]                            # order 1 := order 0: still a valid threaded order, no longer a mirror
    hsc = hip.Scene.from_tables(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials, t)
    assert not hip.set_ordered_tree(hsc, True)
    rays = closest_like_rays(sd, 50000, 33)
    dr = torch.from_numpy(rays).cuda()
    _same(hip.trace_closest(hsc, dr), hip.trace_closest_wave(hsc, dr))
    assert hip.set_ordered_tree(base, True)                  # and the scene those tables came from has them
