"""CPU, build container only: the oracle against the reference's own functions compiled from
/root/reference/src (oracle/_ref/libref_subset.so) and against rocThrust (libthrust_probe.so), on
fresh random inputs -- larger than the committed golden vectors.  Skipped where oracle/_ref is absent."""
import ctypes as C

import numpy as np
import pytest

from oracle import binding as ob
from restir_amd.ctypes_structs import MATERIAL_DTYPE, copy_camera, make_camera
from tests.common import bits_equal

R = ob.ref_subset()
T = ob.thrust_probe()
needs_ref = pytest.mark.skipif(R is None or T is None, reason="oracle/_ref not built (no /root/reference here)")


def unit(rng, n):
    v = rng.normal(size=(n, 3)).astype(np.float32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


@needs_ref
@pytest.mark.parametrize("seed", [1, 2])
def test_triangle_and_box(seed):
    rng = np.random.default_rng(seed)
    n = 100000
    from tests.golden.make_golden import special_rays
    rays = special_rays(rng, n)
    tris = rng.uniform(-1, 1, (n, 9)).astype(np.float32)
    out = []
    for fn in (ob.lib().orc_intersect_triangle, R.ref_intersect_triangle):
        hit = np.zeros(n, np.int32); b = np.zeros((n, 2), np.float32); d = np.zeros(n, np.float32)
        fn(n, rays.reshape(-1), tris.reshape(-1), hit, b.reshape(-1), d)
        out.append((hit, b, d))
    assert np.array_equal(out[0][0], out[1][0])
    m = out[0][0] == 1
    assert bits_equal(out[0][1][m], out[1][1][m]) and bits_equal(out[0][2][m], out[1][2][m])
    boxes = np.sort(rng.uniform(-1.5, 1.5, (n, 2, 3)).astype(np.float32), axis=1).reshape(n, 6).copy()
    res = []
    for fn in (ob.lib().orc_aabb_intersect, R.ref_aabb_intersect):
        hit = np.zeros(n, np.int32); t = np.zeros(n, np.float32)
        fn(n, rays.reshape(-1), boxes.reshape(-1), hit, t)
        res.append((hit, t))
    assert np.array_equal(res[0][0], res[1][0])
    m = res[0][0] == 1
    assert m.sum() > 1000 and bits_equal(res[0][1][m], res[1][1][m])


@needs_ref
def test_rng_against_thrust():
    rng = np.random.default_rng(5)
    seeds = rng.integers(-2 ** 31, 2 ** 31, 20000).astype(np.int32)
    a = np.zeros((len(seeds), 181), np.float32); b = np.zeros_like(a)
    ob.lib().orc_rng_stream_raw(len(seeds), seeds, 181, a.reshape(-1))
    T.thr_rng_stream_raw(len(seeds), seeds, 181, b.reshape(-1))
    assert bits_equal(a, b)


@needs_ref
def test_bsdf_and_math():
    rng = np.random.default_rng(6)
    n = 200000
    mats = np.zeros(n, MATERIAL_DTYPE)
    mats["type"] = rng.integers(0, 5, n); mats["baseColor"] = rng.uniform(0, 1, (n, 3))
    mats["metallic"] = rng.uniform(0, 1, n); mats["roughness"] = rng.uniform(0.02, 1, n); mats["ior"] = 1.5
    nr, wo, wi = unit(rng, n), unit(rng, n), unit(rng, n)
    a = np.zeros((n, 3), np.float32); b = np.zeros_like(a)
    ob.lib().orc_bsdf(n, mats.ctypes.data, nr.reshape(-1), wo.reshape(-1), wi.reshape(-1), a.reshape(-1))
    R.ref_bsdf(n, mats.ctypes.data, nr.reshape(-1), wo.reshape(-1), wi.reshape(-1), b.reshape(-1))
    assert bits_equal(a, b)
    col = rng.uniform(0, 16, (n, 3)).astype(np.float32)
    for mode in (0, 1, 2):
        ob.lib().orc_tonemap(n, col.reshape(-1), mode, a.reshape(-1)); R.ref_tonemap(n, col.reshape(-1), mode, b.reshape(-1))
        assert bits_equal(a, b)


@needs_ref
@pytest.mark.parametrize("nt", [1, 2, 3, 50, 3000, 40000])
def test_bvh_builder(nt):
    rng = np.random.default_rng(nt)
    v = (rng.uniform(-5, 5, (nt, 1, 3)) + rng.uniform(-.3, .3, (nt, 3, 3))).astype(np.float32)
    if nt == 50:
        v[:, :, 2] = 1.0
    ba, na = ob.bvh_build(v); bb, nb = ob.bvh_build(v, R.ref_bvh_build)
    assert bits_equal(ba, bb) and np.array_equal(na, nb)


@needs_ref
def test_camera():
    rng = np.random.default_rng(8)
    for args in [(256, 256, (0, 1, 3.5), (-90, 0, 0), 19.5), (1280, 720, (-3, 4, 9), (200, 25, 0), 35.0)]:
        a = make_camera(*args); b = copy_camera(a)
        ob.camera_update(a); R.ref_camera_update(C.byref(b))
        for f in ("view", "up", "right", "rotationMatInv"):
            assert bits_equal(np.array(getattr(a, f), np.float32), np.array(getattr(b, f), np.float32))
        k = 50000
        xy = np.stack([rng.integers(0, args[0], k), rng.integers(0, args[1], k)], 1).astype(np.int32)
        r4 = rng.uniform(0, 1, (k, 4)).astype(np.float32)
        ra = np.zeros((k, 6), np.float32); rb = np.zeros_like(ra)
        ob.lib().orc_camera_sample(C.byref(b), k, xy.reshape(-1), r4.reshape(-1), ra.reshape(-1))
        R.ref_camera_sample(C.byref(b), k, xy.reshape(-1), r4.reshape(-1), rb.reshape(-1))
        assert bits_equal(ra, rb)


@needs_ref
def test_texture_and_environment_helpers():
    """image.h linearSample (through DevTextureObj) and mathUtil.h toSphere / toPlane / localToWorld, 2e5 inputs each."""
    rng = np.random.default_rng(5)
    tex = rng.uniform(0, 2, (37, 53, 3)).astype(np.float32)
    uv = rng.uniform(-3, 3, (200000, 2)).astype(np.float32)
    uv[:1000] = rng.integers(-2, 3, (1000, 2)).astype(np.float32)
    uv[1000:2000] = (rng.integers(0, 53, (1000, 2)) / np.float32(53)).astype(np.float32)
    b = np.zeros((len(uv), 3), np.float32); R.ref_linear_sample(53, 37, tex.reshape(-1), len(uv), uv.reshape(-1), b.reshape(-1))
    assert bits_equal(ob.linear_sample(tex, uv), b)
    u2 = rng.uniform(0, 1, (200000, 2)).astype(np.float32)
    b = np.zeros((len(u2), 3), np.float32); R.ref_to_sphere(len(u2), u2.reshape(-1), b.reshape(-1))
    assert bits_equal(ob.to_sphere(u2), b)
    d = rng.normal(size=(200000, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:100, 0] = 0; d[100:200, 2] = 0; d[200:300] = [0, 1, 0]
    b = np.zeros((len(d), 2), np.float32); R.ref_to_plane(len(d), d.reshape(-1), b.reshape(-1))
    assert bits_equal(ob.to_plane(d), b)
    n = rng.normal(size=(200000, 3)).astype(np.float32); n /= np.linalg.norm(n, axis=1, keepdims=True); n[:100] = [0, 1, 0]
    v = rng.uniform(-1, 1, (200000, 3)).astype(np.float32)
    b = np.zeros((len(n), 3), np.float32); R.ref_local_to_world(len(n), n.reshape(-1), v.reshape(-1), b.reshape(-1))
    assert bits_equal(ob.local_to_world(n, v), b)


@needs_ref
def test_material_sample_and_pdf():
    """Material::sample / pdf against the reference's material.h on 3e5 random inputs of every type."""
    rng = np.random.default_rng(3)
    n = 300000
    mats = np.zeros(n, MATERIAL_DTYPE)
    mats["type"] = rng.integers(0, 5, n); mats["baseColor"] = rng.uniform(0, 1, (n, 3))
    mats["metallic"] = rng.uniform(0, 1, n); mats["roughness"] = rng.uniform(0.02, 1, n); mats["ior"] = rng.uniform(1.0, 2.5, n)
    mats["metallic"][:1000] = 0; mats["metallic"][1000:2000] = 1; mats["roughness"][2000:3000] = 0
    nrm, wo, wi = unit(rng, n), unit(rng, n), unit(rng, n)
    nrm[:500] = [0, 1, 0]; wo[500:1000] = nrm[500:1000]
    r3 = rng.uniform(0, 1, (n, 3)).astype(np.float32); r3[:50] = 0; r3[50:100] = 1
    a = ob.material_sample(mats, nrm, wo, r3); b = ob.material_sample(mats, nrm, wo, r3, R.ref_material_sample)
    assert np.array_equal(a[3], b[3])
    assert bits_equal(a[0], b[0]) and bits_equal(a[1], b[1]) and bits_equal(a[2], b[2])
    assert bits_equal(ob.material_pdf(mats, nrm, wo, wi), ob.material_pdf(mats, nrm, wo, wi, R.ref_material_pdf))


@needs_ref
def test_closest_hit_loop_over_reference_intersections():
    """bench.py's "reference loop" CPU baseline -- DevScene::intersect's traversal around the reference's compiled
    AABB::intersect / intersectTriangle on the reference builder's tree -- returns the oracle's hits (primitive ids; the oracle
    reports position, not distance, so distances are compared through it)."""
    from tests.common import get_scene, oracle_scene
    sd = get_scene("sponza:0.05")
    W, H = 160, 90
    cam = ob.camera_update(sd.camera(W, H))
    rng = np.random.default_rng(2)
    ys, xs = np.mgrid[0:H, 0:W]
    xy = np.stack([xs.reshape(-1), ys.reshape(-1)], 1).astype(np.int32)
    r4 = rng.uniform(0, 1, (len(xy), 4)).astype(np.float32)
    rays = np.zeros((len(xy), 6), np.float32)
    ob.lib().orc_camera_sample(C.byref(cam), len(xy), xy.reshape(-1), r4.reshape(-1), rays.reshape(-1))
    prim, dist, seconds = ob.ref_closest_hit_loop(R, sd.vertices, rays)
    oprim, _, opos, _, _ = oracle_scene(sd).intersect(rays)
    assert np.array_equal(prim, oprim)
    hit = prim >= 0
    assert hit.mean() > 0.5 and seconds > 0
    d = np.linalg.norm(opos[hit] - rays[hit, :3], axis=1)
    assert np.allclose(d, dist[hit], rtol=1e-4, atol=1e-4)
