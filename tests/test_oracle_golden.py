"""CPU: the oracle (oracle/restir_oracle.c) against the committed golden vectors.

tests/golden/functions_ref.npz was produced by the reference's own functions compiled from
/root/reference/src (oracle/_ref, see tests/golden/make_golden.py) and by rocThrust's RNG; every
comparison is bit-exact.  frames_oracle.npz is a regression pin of the oracle's own frame output.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import binding as ob
from restir_amd.ctypes_structs import Camera, MATERIAL_DTYPE, make_camera
from tests.common import OracleRenderer, bits_equal, get_scene

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLD, "functions_ref.npz"))


def test_intersect_triangle(g):
    n = len(g["tri_hit"])
    hit = np.zeros(n, np.int32); bary = np.zeros((n, 2), np.float32); dist = np.zeros(n, np.float32)
    ob.lib().orc_intersect_triangle(n, g["tri_rays"].reshape(-1), g["tri_tris"].reshape(-1), hit, bary.reshape(-1), dist)
    assert np.array_equal(hit, g["tri_hit"])
    m = hit == 1
    assert m.sum() > 20
    assert bits_equal(bary[m], g["tri_bary"][m]) and bits_equal(dist[m], g["tri_dist"][m])


def test_aabb_intersect_including_special_cases(g):
    n = len(g["box_hit"])
    hit = np.zeros(n, np.int32); t = np.zeros(n, np.float32)
    ob.lib().orc_aabb_intersect(n, g["tri_rays"].reshape(-1), g["box_boxes"].reshape(-1), hit, t)
    assert np.array_equal(hit, g["box_hit"])
    m = hit == 1
    assert m.sum() > 100
    assert bits_equal(t[m], g["box_tmin"][m])


def test_utilhash(g):
    out = np.zeros_like(g["hash_in"])
    ob.lib().orc_utilhash(len(out), g["hash_in"], out)
    assert np.array_equal(out, g["hash_out"])


def test_rng_stream_matches_thrust(g):
    s = np.zeros_like(g["rng_stream"])
    ob.lib().orc_rng_stream_raw(s.shape[0], g["rng_seeds"], s.shape[1], s.reshape(-1))
    assert bits_equal(s, g["rng_stream"])
    assert s.min() >= 0.0 and s.max() <= 1.0


def test_bsdf(g):
    m = len(g["bsdf_out"])
    mats = np.ascontiguousarray(g["bsdf_mats"]).view(MATERIAL_DTYPE).reshape(m)
    out = np.zeros((m, 3), np.float32)
    ob.lib().orc_bsdf(m, mats.ctypes.data, g["bsdf_n"].reshape(-1), g["bsdf_wo"].reshape(-1), g["bsdf_wi"].reshape(-1), out.reshape(-1))
    assert bits_equal(out, g["bsdf_out"])


@pytest.mark.parametrize("i", [0, 1])
def test_camera(g, i):
    a = g[f"cam{i}_args"]
    cam = make_camera(int(a[0]), int(a[1]), a[2:5], a[5:8], float(a[8]))
    ob.camera_update(cam)
    ref = Camera.from_buffer_copy(g[f"cam{i}_struct"].tobytes())
    for f in ("view", "up", "right", "rotationMatInv"):
        assert bits_equal(np.array(getattr(cam, f), np.float32), np.array(getattr(ref, f), np.float32)), f
    k = len(g[f"cam{i}_xy"])
    rays = np.zeros((k, 6), np.float32)
    ob.lib().orc_camera_sample(C.byref(cam), k, g[f"cam{i}_xy"].reshape(-1), g[f"cam{i}_r4"].reshape(-1), rays.reshape(-1))
    assert bits_equal(rays, g[f"cam{i}_rays"])
    pos = np.zeros((k, 3), np.float32)
    ob.lib().orc_camera_position(C.byref(cam), k, g[f"cam{i}_xy"].reshape(-1), g[f"cam{i}_dist"], pos.reshape(-1))
    assert bits_equal(pos, g[f"cam{i}_pos"])
    rc = np.zeros((k, 2), np.int32)
    ob.lib().orc_camera_raster_coord(C.byref(cam), k, pos.reshape(-1), rc.reshape(-1))
    assert np.array_equal(rc, g[f"cam{i}_raster"])


def test_math_helpers(g):
    n = len(g["ruv"])
    o = np.zeros((n, 3), np.float32)
    ob.lib().orc_sample_triangle_uniform(n, g["tri_tris"].reshape(-1), g["ruv"].reshape(-1), o.reshape(-1))
    assert bits_equal(o, g["sample_tri"])
    d = np.zeros((n, 2), np.float32)
    ob.lib().orc_to_concentric_disk(n, g["ruv"].reshape(-1), d.reshape(-1))
    assert bits_equal(d, g["disk"])
    area = np.zeros(n, np.float32); nrm = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
    ob.lib().orc_triangle_misc(n, g["tri_tris"].reshape(-1), g["misc_x"].reshape(-1), area, nrm.reshape(-1), pdf)
    assert bits_equal(area, g["tri_area"]) and bits_equal(nrm, g["tri_normal"]) and bits_equal(pdf, g["tri_pdf"])


def test_texture_and_environment_helpers(g):
    """image.h linearSample and mathUtil.h toSphere / toPlane / localToWorld against the reference's own code
    (default libm mode = glibc, which is what the reference's host-compiled functions call)."""
    assert bits_equal(ob.linear_sample(g["tex"], g["tex_uv"]), g["tex_sample"])
    assert bits_equal(ob.to_sphere(g["sph_uv"]), g["sph_dir"])
    assert bits_equal(ob.to_plane(g["plane_dir"]), g["plane_uv"])
    assert bits_equal(ob.local_to_world(g["l2w_n"], g["l2w_v"]), g["l2w_out"])
    # the correctly rounded mode differs from glibc by at most one ulp, and only for a small fraction of arguments
    ob.set_libm_mode(1)
    try:
        d = ob.to_sphere(g["sph_uv"])
    finally:
        ob.set_libm_mode(0)
    ulp = np.abs(d.view(np.int32).astype(np.int64) - g["sph_dir"].view(np.int32).astype(np.int64))
    assert ulp.max() <= 4 and (ulp > 0).mean() < 0.1


def test_material_sample_and_pdf(g):
    """Material::sample / pdf (material.h:230-256: cosine hemisphere, GGX visible normals, dielectric reflect / refract)."""
    d, b, p, t = ob.material_sample(g["ms_mats"], g["ms_n"], g["ms_wo"], g["ms_r"])
    assert np.array_equal(t, g["ms_type"])
    assert bits_equal(d, g["ms_dir"]) and bits_equal(b, g["ms_bsdf"]) and bits_equal(p, g["ms_pdf"])
    assert bits_equal(ob.material_pdf(g["ms_mats"], g["ms_n"], g["ms_wo"], g["ms_wi"]), g["ms_pdf_eval"])
    assert set(np.unique(t)) >= {1 | 16, 2 | 16, 4 | 16, 4 | 32, 1 << 15}      # diffuse, glossy, specular R / T, invalid


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_tonemap(g, mode):
    o = np.zeros_like(g["tonemap_in"])
    ob.lib().orc_tonemap(len(o), g["tonemap_in"].reshape(-1), mode, o.reshape(-1))
    assert bits_equal(o, g[f"tonemap{mode}"])


@pytest.mark.parametrize("name", ["a", "b"])
def test_bvh_build(g, name):
    boxes, nodes = ob.bvh_build(g[f"bvh_{name}_verts"])
    assert bits_equal(boxes, g[f"bvh_{name}_boxes"])
    assert np.array_equal(nodes, g[f"bvh_{name}_nodes"])


def test_frames_regression_pin():
    """Oracle frame output is unchanged since the fixtures were generated (not reference-derived)."""
    fr = np.load(os.path.join(GOLD, "frames_oracle.npz"))
    sd = get_scene("cornell")
    for reuse in (0, 1, 2, 3):
        o = OracleRenderer(sd, 64, 64)
        for _ in range(3):
            img = o.frame(reuse)
        assert bits_equal(img, fr[f"cornell64_reuse{reuse}_frame2"]), reuse
        assert np.array_equal(o.restir.last["numSamples"], fr[f"cornell64_reuse{reuse}_M"])
    o = OracleRenderer(sd, 64, 64)
    assert bits_equal(o.frame(0, use_reservoir=False), fr["cornell64_ptdirect"])
    o = OracleRenderer(sd, 48, 48)
    d = np.zeros((48 * 48, 3), np.float32); i1 = np.zeros_like(d); i2 = np.zeros_like(d); i3 = np.zeros_like(d)
    ob.path_trace(o.scene, o.cam, d, i1, 0, 0, 4)
    ob.pt_indirect(o.scene, o.cam, i2, 0, 1, 4)
    for frame in range(3):
        o.gbuf.render(o.scene, o.cam); o.restir.indirect(o.scene, o.cam, o.gbuf, i3, 0, frame, 1, 4); o.gbuf.update(o.cam)
    assert bits_equal(d, fr["cornell48_pt_direct"]) and bits_equal(i1, fr["cornell48_pt_indirect"])
    assert bits_equal(i2, fr["cornell48_ptind"]) and bits_equal(i3, fr["cornell48_gi"])
    assert np.array_equal(o.restir.ind_last["numSamples"], fr["cornell48_gi_M"])
    for name in ("cornell_textured", "cornell_maps"):
        for mode in (0, 1):
            ob.set_libm_mode(mode)
            try:
                o = OracleRenderer(get_scene(name), 64, 48)
                for _ in range(2):
                    img = o.frame(3)
            finally:
                ob.set_libm_mode(0)
            assert bits_equal(img, fr[f"{name}_libm{mode}_frame1"]), (name, mode)
            assert bits_equal(o.gbuf.albedo, fr[f"{name}_libm{mode}_albedo"]), (name, mode)


def test_svgf_oracle_properties():
    """SpatioTemporalFilter on the oracle: a static camera accumulates history (moment count grows, variance of a
    noisy input falls), and misses / light pixels pass through unfiltered."""
    sd = get_scene("cornell")
    W, H = 64, 64
    o = OracleRenderer(sd, W, H)
    f = ob.SVGF(W, H)
    var = []
    for frame in range(6):
        o.gbuf.render(o.scene, o.cam)
        o.restir.direct(o.scene, o.cam, o.gbuf, o.image, 0, o.looper, 0)
        o.looper += 1
        out = f.filter(o.image, o.gbuf, o.cam)
        st = f.state()
        shaded = o.gbuf.prim_id[o.gbuf.frame_idx] > -1
        assert np.isfinite(out).all()
        assert np.array_equal(out[~shaded], o.image[~shaded])              # primId <= NullPrimitive: copied
        assert st["accum_moment"][shaded, 2].max() == float(frame)
        var.append(float(st["variance"][shaded].mean()))
        f.next_frame()
        o.gbuf.update(o.cam)
    assert var[-1] < var[0]
    noisy = np.abs(o.image[shaded] - out[shaded]).mean()
    assert noisy > 1e-3


def test_config1_cornell_256_host_loop():
    """BASELINE config 1: Cornell box, 256x256, 1 spp raw direct path trace, looper 0 (host loop)."""
    sd = get_scene("cornell")
    o = OracleRenderer(sd, 256, 256)
    img = o.frame(0, use_reservoir=False)
    assert o.rays >= 256 * 256 and np.isfinite(img).all()
    lum = img.mean()
    assert 0.2 < lum < 0.6
    # the light is visible: some pixels equal its radiance exactly
    assert (img == 10.0).all(axis=1).sum() > 50
