"""GPU parity tests: librestir_hip (through the C ABI) against the CPU oracle on the same seeded
inputs.  Tolerances (BASELINE.json north_star: per-pixel L1 radiance error < 1e-4):

  * integer planes (G-buffer ids, motion, reservoir M, hit primitive ids): exact;
  * FP32 planes produced only by + - * / sqrt (rays, hits, G-buffer, RIS / temporal reservoirs,
    radiance without spatial reuse): BIT-EXACT -- device code is built with -ffp-contract=off and
    IEEE divide/sqrt, the oracle with the same op order;
  * spatial reuse (uses sinf/cosf for the tap position) : mean per-pixel L1 < 1e-4 and
    fraction of pixels with L1 > 1e-3 below 1e-3 (expected: 0 mismatches);
  * tone-mapped RGBA8 (powf): at most 1 LSB on at most 1e-4 of the bytes;
  * EAW (expf): relative 1e-5.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import binding as ob
from restir_amd.ctypes_structs import RESERVOIR_DTYPE
from tests.common import (HipRenderer, OracleRenderer, bits_equal, get_scene, hip_scene, mismatch_fraction, oracle_scene,
                          radiance_stats)

pytestmark = pytest.mark.gpu

SCENES = {"cornell": (128, 128), "sponza:0.03": (160, 96)}


def _random_rays(sd, n, seed):
    rng = np.random.default_rng(seed)
    v = sd.vertices.reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    o = rng.uniform(lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo), (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    k = n // 16                              # axis-aligned and near-zero-component directions (bvh.h:91-146)
    d[:k] = 0; d[np.arange(k), rng.integers(0, 3, k)] = rng.choice([-1.0, 1.0], k)
    d[k:2 * k, 0] = rng.uniform(-1e-6, 1e-6, k)
    d[2 * k:3 * k, 1] = 0.0
    nn = np.linalg.norm(d, axis=1, keepdims=True)
    d = (d / nn).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([o, d], 1), np.float32)


@pytest.mark.parametrize("name", list(SCENES))
def test_trace_closest_and_occlusion(hip, name):
    import torch
    sd = get_scene(name)
    osc = oracle_scene(sd)
    hsc = hip.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
    rays = _random_rays(sd, 40000, 3)
    prim, mat, pos, nrm, _ = osc.intersect(rays)
    gp, gm, gpos, gn = hip.trace_closest(hsc, torch.from_numpy(rays).cuda())
    assert np.array_equal(prim, gp.cpu().numpy())
    hitmask = prim >= 0
    assert hitmask.sum() > 1000
    assert np.array_equal(mat[hitmask], gm.cpu().numpy()[hitmask])
    assert bits_equal(pos[hitmask], gpos.cpu().numpy()[hitmask])
    assert bits_equal(nrm[hitmask], gn.cpu().numpy()[hitmask])
    # occlusion: segments between hit points and random points
    rng = np.random.default_rng(4)
    a = pos[hitmask][:15000]
    b = a[rng.permutation(len(a))]
    seg = np.ascontiguousarray(np.concatenate([a, b], 1), np.float32)
    seg[:50, 3:] = seg[:50, :3]                       # degenerate x == y (Q9): NaN direction must terminate
    occ = osc.test_occlusion(seg)
    gocc = hip.trace_occlusion(hsc, torch.from_numpy(seg).cuda()).cpu().numpy()
    assert np.array_equal(occ, gocc)
    assert 0 < occ.sum() < len(occ)


def _shadow_like_segments(sd, n, seed):
    """Segments of the kinds the shadow pass produces and the corner cases of the occlusion tree: surface point ->
    light point, random pairs, short, grazing along a triangle's plane, axis-aligned / near-zero directions
    (reference walk), and origins far outside the scene (beyond the grid test's reach)."""
    rng = np.random.default_rng(seed)
    v = sd.vertices.reshape(-1, 3, 3).astype(np.float64)
    lights = np.nonzero(sd.materials["type"][sd.material_ids] == 4)[0]

    def points_on(tris, k):
        t = v[rng.choice(tris, k)]
        u = rng.uniform(size=(k, 2)); flip = u.sum(1) > 1; u[flip] = 1 - u[flip]
        p = t[:, 0] * (1 - u.sum(1))[:, None] + t[:, 1] * u[:, :1] + t[:, 2] * u[:, 1:]
        nrm = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
        return p, nrm, t

    a, na, _ = points_on(np.arange(len(v)), n)
    b, _, _ = points_on(lights if len(lights) else np.arange(len(v)), n)
    na *= np.sign(np.einsum("ij,ij->i", na, b - a))[:, None]
    a = a + 1e-4 * na
    k = n // 8
    b[k:2 * k] = points_on(np.arange(len(v)), k)[0]                                   # random surface pairs
    b[2 * k:3 * k] = a[2 * k:3 * k] + rng.normal(size=(k, 3)) * 0.05                      # short
    g, ng, tg = points_on(np.arange(len(v)), k)                                           # grazing: inside a triangle's plane
    a[3 * k:4 * k] = g + (tg[:, 0] - g) * 3.0; b[3 * k:4 * k] = g + (tg[:, 1] - g) * 3.0
    ax = rng.integers(0, 3, k)                                                            # axis-aligned / near-zero components
    b[4 * k:5 * k] = a[4 * k:5 * k]; b[np.arange(4 * k, 5 * k), ax] += rng.choice([-3.0, 3.0], k)
    b[4 * k:4 * k + k // 2, (ax[:k // 2] + 1) % 3] += rng.uniform(-2e-6, 2e-6, k // 2)
    lo, hi = v.reshape(-1, 3).min(0), v.reshape(-1, 3).max(0)
    a[5 * k:5 * k + 64] = hi + (hi - lo) * rng.uniform(6, 40, (64, 1))                    # far origins
    seg = np.ascontiguousarray(np.concatenate([a, b], 1), np.float32)
    seg[-16:, 3:] = seg[-16:, :3]                                                         # x == y: NaN direction
    return seg


def test_occlusion_tree_equals_reference_walk(hip, monkeypatch):
    """Shadow rays go through a second tree + ancestor-chain verification (rs_scene.h walk_occlusion_tree); the
    result must equal DevScene::testOcclusion on the reference's tree for every segment: against the oracle, and at
    a larger count against the library's own reference walk (RS_NO_OCCLUSION_TREE)."""
    import torch
    sd = get_scene("sponza:0.1")
    seg = _shadow_like_segments(sd, 400000, 11)
    fast = hip.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
    monkeypatch.setenv("RS_NO_OCCLUSION_TREE", "1")
    slow = hip.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
    monkeypatch.delenv("RS_NO_OCCLUSION_TREE")
    dseg = torch.from_numpy(seg).cuda()
    a = hip.trace_occlusion(fast, dseg).cpu().numpy()
    b = hip.trace_occlusion(slow, dseg).cpu().numpy()
    assert np.array_equal(a, b)
    assert 0.05 < a.mean() < 0.95
    sub = seg[:: 10]
    assert np.array_equal(oracle_scene(sd).test_occlusion(sub), a[:: 10])


def test_occlusion_tree_with_caller_supplied_boxes(hip, monkeypatch):
    """rs_scene_create accepts any box table.  Boxes that are not nested in their parents switch the leaf
    shortcut off (every ancestor is then tested); results must still equal the reference walk on the same table."""
    import torch
    sd = get_scene("sponza:0.03")
    base = hip.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
    t = base.host_desc()
    rng = np.random.default_rng(5)
    boxes = t["boxes"].copy()
    pick = rng.choice(len(boxes), len(boxes) // 3, replace=False)
    boxes[pick, :3] -= rng.uniform(0, 0.2, (len(pick), 3)).astype(np.float32)      # grow some boxes beyond their parents
    boxes[pick, 3:] += rng.uniform(0, 0.2, (len(pick), 3)).astype(np.float32)
    shrink = rng.choice(len(boxes), len(boxes) // 10, replace=False)                 # and make some inner boxes miss their content
    mid = 0.5 * (boxes[shrink, :3] + boxes[shrink, 3:])
    boxes[shrink, :3] = 0.5 * (boxes[shrink, :3] + mid); boxes[shrink, 3:] = 0.5 * (boxes[shrink, 3:] + mid)
    t["boxes"] = boxes
    args = (sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials, t)
    fast = hip.Scene.from_tables(*args)
    monkeypatch.setenv("RS_NO_OCCLUSION_TREE", "1")
    slow = hip.Scene.from_tables(*args)
    monkeypatch.delenv("RS_NO_OCCLUSION_TREE")
    segh = _shadow_like_segments(sd, 200000, 12)
    seg = torch.from_numpy(segh).cuda()
    a = hip.trace_occlusion(fast, seg).cpu().numpy()
    b = hip.trace_occlusion(slow, seg).cpu().numpy()
    assert np.array_equal(a, b)
    assert 0.02 < a.mean() < 0.98
    # and the literal reference walk on the same table (also covers: the near-zero-direction cull is off for such tables)
    osc = ob.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials,
                   prebuilt=(t["light_prim_ids"], t["light_radiance"], np.zeros(len(t["light_prim_ids"]), np.float32),
                             t["light_prob"], t["light_fail"], t["sum_power"], boxes, t["nodes"]))
    assert np.array_equal(osc.test_occlusion(segh[::20]), a[::20])
    rays = _random_rays(sd, 20000, 6)
    prim = osc.intersect(rays)[0]
    assert np.array_equal(prim, hip.trace_closest(fast, torch.from_numpy(rays).cuda())[0].cpu().numpy())


@pytest.mark.parametrize("name", list(SCENES))
def test_gbuffer(hip, name):
    sd = get_scene(name)
    W, H = SCENES[name]
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    for frame in range(2):                            # second frame exercises motion vectors vs lastCamera
        if frame == 1:
            p = (sd.camera_args["position"][0] + 0.15, sd.camera_args["position"][1], sd.camera_args["position"][2] - 0.1)
            o.set_camera_position(p); h.set_camera_position(p)
        o.gbuf.render(o.scene, o.cam); h.gbuf.render(h.scene, h.cam)
        g = h.gbuf.download()
        f = g["frame_idx"]
        assert f == o.gbuf.frame_idx
        assert np.array_equal(o.gbuf.prim_id[f], g["prim_id"][f])
        assert np.array_equal(o.gbuf.motion, g["motion"])
        assert bits_equal(o.gbuf.albedo, g["albedo"])
        assert bits_equal(o.gbuf.normal[f], g["normal"][f])
        assert bits_equal(o.gbuf.depth[f], g["depth"][f])
        o.gbuf.update(o.cam); h.gbuf.update(h.cam)


def _compare_reservoirs(a, b):
    for k in ("numSamples",):
        assert np.array_equal(a[k], b[k]), k
    for k in ("Li", "wi", "dist", "weight"):
        assert bits_equal(a[k], b[k]), k


@pytest.mark.parametrize("name", list(SCENES))
@pytest.mark.parametrize("reuse", [0, 1])
@pytest.mark.parametrize("table", ["lds", "global"])
def test_restir_no_spatial_bit_exact(hip, name, reuse, table):
    """RIS-only (config 2 semantics) and temporal reuse: bit-exact radiance and reservoirs over 4 frames, with the RIS light table
    in LDS (what a full frame uses) and read from global memory (what a launch below 384 Ki pixels uses by default)."""
    sd = get_scene(name)
    W, H = SCENES[name]
    o = OracleRenderer(sd, W, H)
    hip.set_ris_table_pixels(0 if table == "lds" else 1 << 30)
    try:
        h = HipRenderer(hip, sd, W, H)
        for frame in range(4):
            a = o.frame(reuse); b = h.frame(reuse)
            assert o.rays == h.rays, (frame, o.rays, h.rays)
            assert bits_equal(a, b), (frame, radiance_stats(a, b))
            _compare_reservoirs(o.restir.last, h.restir.download(1))      # the buffer written this frame
    finally:
        hip.set_ris_table_pixels(384 * 1024)


@pytest.mark.parametrize("name", list(SCENES))
@pytest.mark.parametrize("reuse", [2, 3])
@pytest.mark.parametrize("libm", ["correctly_rounded", "glibc"])
def test_restir_spatial(hip, name, reuse, libm):
    """Spatial reuse.  With the oracle's cos / sin correctly rounded (what the device evaluates) every bit
    of the radiance and of both reservoir buffers must agree; against glibc's cosf / sinf (one ulp off for 1.3 % of the
    arguments, which moves a tap only when that ulp crosses a pixel boundary) the stated tolerance applies."""
    sd = get_scene(name)
    W, H = SCENES[name]
    ob.set_libm_mode(1 if libm == "correctly_rounded" else 0)
    try:
        o = OracleRenderer(sd, W, H)
        h = HipRenderer(hip, sd, W, H)
        for frame in range(4):
            a = o.frame(reuse); b = h.frame(reuse)
            st = radiance_stats(a, b)
            if libm == "correctly_rounded":
                assert st["bit_mismatch"] == 0, (frame, st)
            assert st["mean_l1"] < 1e-4 and st["flip_frac"] <= 1e-3, (frame, st)
            _compare_reservoirs(o.restir.last, h.restir.download(1))
            _compare_reservoirs(o.restir.temp, h.restir.download(2))
    finally:
        ob.set_libm_mode(0)


def test_spatial_pass_with_bands_of_unequal_width(hip):
    """k_spatial_shade sweeps frames wider than 48 tiles in vertical bands (restir.hip, kBandTiles).  Every size of the other tests divides
    into equal bands; 2016 pixels are 63 tiles = bands of 32 and 31, and 1570 pixels 50 tiles (the last one 2 pixels wide) = 25 + 25 --
    the narrower last band and the partial tile must neither drop nor duplicate a tile: radiance and reservoirs against the oracle."""
    ob.set_libm_mode(1)
    try:
        for W, H in ((2016, 40), (1570, 35)):
            sd = get_scene("cornell")
            o = OracleRenderer(sd, W, H)
            h = HipRenderer(hip, sd, W, H)
            for frame in range(3):
                a = o.frame(3); b = h.frame(3)
                st = radiance_stats(a, b)
                assert st["bit_mismatch"] == 0, (W, H, frame, st)
                _compare_reservoirs(o.restir.last, h.restir.download(1))
            assert np.abs(a).sum() > 0
    finally:
        ob.set_libm_mode(0)


def test_tap_estimate_error_is_inside_the_fallback_band(hip):
    """The spatial pass estimates tap positions with v_sqrt/v_sin/v_cos and re-evaluates exactly when
    an integer lies within kTapErr (4e-5) of the estimate (restir.hip disk_tap).  The band must cover
    the hardware error with margin."""
    err = C.c_float(0)
    hip.check(hip.lib().rs_debug_tap_estimate_error(1 << 24, C.byref(err)))
    assert 0.0 < err.value < 1e-5, err.value          # 4x margin to kTapErr


def test_sqrt_of_uniform_equals_sqrtf_on_every_variate(hip):
    """The light sampler's square root of a uniform variate skips the range scaling and class test of the exactly rounded
    expansion (rs_surface.h sqrt_of_uniform): all 2^31 - 2 values Rng::uniform() can return must give sqrtf's bits."""
    bad = C.c_ulonglong(1)
    hip.check(hip.lib().rs_debug_sqrt_of_uniform_mismatches(C.byref(bad)))
    assert bad.value == 0, bad.value


def test_spatial_many_frames_large(hip):
    """A larger frame and more frames for the tap-position fast path: every pixel must still agree."""
    sd = get_scene("sponza:0.03")
    W, H = 480, 270
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    for frame in range(3):
        a = o.frame(3); b = h.frame(3)
        assert bits_equal(a, b), (frame, radiance_stats(a, b))


def test_config2_cornell_720p_ris_only(hip):
    """BASELINE config 2: Cornell box + 1 area light, 1280x720, RIS only (32 candidates), frames 0-7 -- bit-exact."""
    sd = get_scene("cornell")
    W, H = 1280, 720
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    for frame in range(8):
        a = o.frame(0); b = h.frame(0)
        assert o.rays == h.rays, (frame, o.rays, h.rays)
        assert bits_equal(a, b), (frame, radiance_stats(a, b))


def test_many_lights_use_the_global_memory_ris_kernel(hip):
    """Bistro-class scene with 1228 emissive triangles (> 1024: k_ris reads the light table from global memory
    instead of the LDS copy), spatiotemporal reuse, bit-exact."""
    sd = get_scene("bistro:0.12")
    assert int((sd.materials["type"][sd.material_ids] == 4).sum()) > 1024
    W, H = 160, 96
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    for frame in range(3):
        a = o.frame(3); b = h.frame(3)
        assert o.rays == h.rays
        assert bits_equal(a, b), (frame, radiance_stats(a, b))
        _compare_reservoirs(o.restir.last, h.restir.download(1))


def _edge_scene(kind):
    from restir_amd import scenes
    from restir_amd.ctypes_structs import LAMBERTIAN
    sd = get_scene("cornell")
    if kind == "no_lights":                         # lightSampler.length == 0: every candidate pdf is -1, W stays 0
        sd.materials = sd.materials.copy(); sd.materials[3]["type"] = LAMBERTIAN
    elif kind == "dielectric_disney":               # Material::BSDF returns 0 for both; the dielectric keeps its normal unflipped
        sd.materials = sd.materials.copy(); sd.materials[0]["type"] = 2; sd.materials[1]["type"] = 3
    elif kind == "one_triangle":                    # BVHSize 1: the root is a leaf
        sd.vertices = sd.vertices[:1].copy(); sd.normals = sd.normals[:1].copy(); sd.texcoords = sd.texcoords[:1].copy()
        sd.material_ids = sd.material_ids[:1].copy()
    elif kind == "two_triangles_light_only":        # the camera sees only the light and the void
        sd.vertices = sd.vertices[-2:].copy(); sd.normals = sd.normals[-2:].copy(); sd.texcoords = sd.texcoords[-2:].copy()
        sd.material_ids = sd.material_ids[-2:].copy()
    return sd


@pytest.mark.parametrize("kind,size", [("plain", (97, 61)), ("plain", (33, 9)), ("no_lights", (64, 48)), ("dielectric_disney", (64, 48)),
                                       ("one_triangle", (40, 24)), ("two_triangles_light_only", (40, 24))])
def test_edge_cases_bit_exact(hip, kind, size):
    """Ragged frame sizes (partial 32x8 / 32x16 tiles, spatial taps clipped at every border), a scene without lights,
    BSDF types that evaluate to zero, and degenerate BVH sizes: G-buffer, radiance and reservoirs bit for bit."""
    sd = _edge_scene(kind)
    W, H = size
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    for frame in range(3):
        a = o.frame(3); b = h.frame(3)
        assert o.rays == h.rays, (frame, o.rays, h.rays)
        assert bits_equal(a, b), (frame, radiance_stats(a, b))
        _compare_reservoirs(o.restir.last, h.restir.download(1))
    g = h.gbuf.download(); f = g["frame_idx"] ^ 1
    assert np.array_equal(o.gbuf.prim_id[f], g["prim_id"][f]) and bits_equal(o.gbuf.depth[f], g["depth"][f])
    assert np.array_equal(o.gbuf.motion, g["motion"])
    a = OracleRenderer(sd, W, H).frame(0, use_reservoir=False); b = HipRenderer(hip, sd, W, H).frame(0, use_reservoir=False)
    assert bits_equal(a, b)


@pytest.fixture
def correctly_rounded_libm():
    """The environment-map and procedural-texture paths call sin / cos / atan2; librestir_hip evaluates them
    correctly rounded (rs_surface.h), so the bit-exact comparisons run the oracle in the same mode."""
    ob.set_libm_mode(1)
    yield
    ob.set_libm_mode(0)


@pytest.mark.parametrize("name", ["cornell_textured", "cornell_maps"])
def test_textures_and_environment_map_bit_exact(hip, correctly_rounded_libm, name):
    """getTexturedMaterialAndSurface (base colour / procedural / metallic / roughness / normal maps) and the
    environment map as miss radiance, G-buffer albedo and last light of the sampler (src/scene.h:78-99,358-403):
    G-buffer planes, ReSTIR radiance and reservoirs in all reuse modes, and the PT-direct baseline, bit for bit."""
    sd = get_scene(name)
    W, H = 160, 120
    for reuse in (0, 3):
        o = OracleRenderer(sd, W, H)
        h = HipRenderer(hip, sd, W, H)
        for frame in range(3):
            a = o.frame(reuse); b = h.frame(reuse)
            assert o.rays == h.rays, (reuse, frame, o.rays, h.rays)
            assert bits_equal(a, b), (reuse, frame, radiance_stats(a, b))
            _compare_reservoirs(o.restir.last, h.restir.download(1))
        g = h.gbuf.download()
        f = g["frame_idx"] ^ 1                                       # the planes rendered last (update() flipped the index)
        assert np.array_equal(o.gbuf.prim_id[f], g["prim_id"][f])
        assert bits_equal(o.gbuf.albedo, g["albedo"])
        assert bits_equal(o.gbuf.normal[f], g["normal"][f])
        if sd.env_map_tex >= 0:
            miss = o.gbuf.prim_id[f] == -1
            assert miss.sum() > 200 and np.all(o.gbuf.albedo[miss].max(1) > 0)      # environment radiance in the albedo plane
    o = OracleRenderer(sd, W, H); h = HipRenderer(hip, sd, W, H)
    a = o.frame(0, use_reservoir=False); b = h.frame(0, use_reservoir=False)
    assert o.rays == h.rays and bits_equal(a, b), radiance_stats(a, b)
    # the host build of the light sampler with the environment map as its last entry (scene.cpp:136-157)
    t = h.scene.host_desc()
    assert bits_equal(t["light_prob"], o.scene.light_prob) and np.array_equal(t["light_fail"], o.scene.light_fail)
    assert t["sum_power"] == o.scene.sum_power and t["env_map_tex"] == sd.env_map_tex
    assert bits_equal(t["env_prob"], o.scene.env_prob) and np.array_equal(t["env_fail"], o.scene.env_fail)


def _gi_scene(name):
    if name == "cornell_glass":                     # a dielectric box (reflect / refract / total internal reflection) and a Disney wall (never sampled)
        sd = get_scene("cornell")
        sd.materials = sd.materials.copy()
        sd.materials = np.concatenate([sd.materials, sd.materials[:1]])
        sd.materials[4]["type"] = 2; sd.materials[4]["ior"] = 1.5; sd.materials[4]["baseColor"] = (0.9, 0.95, 1.0)
        sd.material_ids = sd.material_ids.copy(); sd.material_ids[22:34] = 4
        sd.materials[2]["type"] = 3
        return sd
    return get_scene(name)


@pytest.mark.parametrize("name", ["cornell", "cornell_glass", "cornell_textured", "sponza:0.03"])
def test_multi_bounce_kernels_bit_exact(hip, correctly_rounded_libm, name):
    """pathTrace (singleKernelPT), pathTraceIndirect (PTIndirectKernel) and ReSTIRIndirect (ReSTIRIndirectKernel) with
    Material::sample / pdf for every BSDF type: images, ray counts and the 68-byte indirect reservoirs, bit for bit
    over several frames and trace depths (the BSDF sampling's cos / sin are correctly rounded on both sides)."""
    import torch
    sd = _gi_scene(name)
    W, H = 96, 64
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    od = np.zeros((W * H, 3), np.float32); oi = np.zeros((W * H, 3), np.float32)
    hd = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda"); hi = torch.zeros_like(hd)
    for frame, depth in enumerate((1, 3, 5)):                           # iter accumulates like Settings::accumulate
        ra = ob.path_trace(o.scene, o.cam, od, oi, frame, frame, depth)
        rb = hip.path_trace(h.scene, h.cam, hd.data_ptr(), hi.data_ptr(), frame, frame, depth)
        assert ra == rb, (frame, ra, rb)
        assert bits_equal(od, hd.cpu().numpy()) and bits_equal(oi, hi.cpu().numpy()), (frame, radiance_stats(oi, hi.cpu().numpy()))
    assert oi.max() > 0
    oi[:] = 0; hi.zero_()
    for frame, depth in enumerate((2, 4)):
        ra = ob.pt_indirect(o.scene, o.cam, oi, frame, 7 + frame, depth)
        rb = hip.path_trace_indirect(h.scene, h.cam, hi.data_ptr(), frame, 7 + frame, depth)
        assert ra == rb and bits_equal(oi, hi.cpu().numpy()), (frame, radiance_stats(oi, hi.cpu().numpy()))
    # ReSTIR-GI over a moving camera: G-buffer reprojection feeds findTemporalNeighbor
    from restir_amd.scenes import orbit_position
    oi[:] = 0; hi.zero_()
    for frame in range(4):
        p = orbit_position(sd.camera_args["position"], frame, radius=0.2)
        o.set_camera_position(p); h.set_camera_position(p)
        o.gbuf.render(o.scene, o.cam); h.gbuf.render(h.scene, h.cam)
        ra = o.restir.indirect(o.scene, o.cam, o.gbuf, oi, 0, frame, 1, 4)
        rb = h.restir.indirect(h.scene, h.cam, h.gbuf, hi.data_ptr(), 0, frame, 1, 4)
        assert ra == rb, (frame, ra, rb)
        assert bits_equal(oi, hi.cpu().numpy()), (frame, radiance_stats(oi, hi.cpu().numpy()))
        a, b = o.restir.ind_last, h.restir.download_indirect(1)
        assert np.array_equal(a["numSamples"], b["numSamples"])
        for k in ("Lo", "xv", "nv", "xs", "ns", "weight"):
            assert bits_equal(a[k], b[k]), (frame, k)
        o.gbuf.update(o.cam); h.gbuf.update(h.cam)
    assert o.restir.ind_last["numSamples"].max() > 2 and oi.max() > 0


@pytest.mark.parametrize("name", ["cornell", "cornell_textured", "sponza:0.03"])
def test_restir_indirect_shallow_depths(hip, correctly_rounded_libm, name):
    """ReSTIRIndirect at trace depths 1 and 2.  The last bounce of a path only asks whether its closest hit is emissive and is answered
    through the emissive triangles' own tree (rs_scene.h may_hit_emissive_wave) -- except at depth 1 of ReSTIRIndirect, whose sample records
    the hit point whatever it is (src/restir.cu:345-360), and in scenes with an environment map, where a miss contributes: both exceptions
    and the rule itself against the oracle, images, ray counts and reservoirs bit for bit."""
    import torch
    sd = _gi_scene(name)
    W, H = 96, 64
    for depth in (1, 2):
        o = OracleRenderer(sd, W, H)
        h = HipRenderer(hip, sd, W, H)
        oi = np.zeros((W * H, 3), np.float32)
        hi = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda")
        for frame in range(2):
            o.gbuf.render(o.scene, o.cam); h.gbuf.render(h.scene, h.cam)
            ra = o.restir.indirect(o.scene, o.cam, o.gbuf, oi, 0, frame, 1, depth)
            rb = h.restir.indirect(h.scene, h.cam, h.gbuf, hi.data_ptr(), 0, frame, 1, depth)
            assert ra == rb, (depth, frame, ra, rb)
            assert bits_equal(oi, hi.cpu().numpy()), (depth, frame, radiance_stats(oi, hi.cpu().numpy()))
            a, b = o.restir.ind_last, h.restir.download_indirect(1)
            assert np.array_equal(a["numSamples"], b["numSamples"])
            for k in ("Lo", "xv", "nv", "xs", "ns", "weight"):
                assert bits_equal(a[k], b[k]), (depth, frame, k)
            o.gbuf.update(o.cam); h.gbuf.update(h.cam)


def test_textured_scene_against_glibc_libm(hip):
    """The same scene against the oracle's default libm mode (glibc sinf / cosf / atan2f, what a host build of the
    reference computes): one-ulp differences of the four libm calls stay inside the stated tolerance."""
    sd = get_scene("cornell_textured")
    W, H = 160, 120
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    for frame in range(3):
        st = radiance_stats(o.frame(3), h.frame(3))
        assert st["mean_l1"] < 1e-4 and st["flip_frac"] <= 1e-3, (frame, st)


def test_restir_moving_camera_temporal(hip, correctly_rounded_libm):
    """Orbiting camera (runCuda :149-153 with a fixed dt): reprojection through devMotion; every bit of every frame."""
    from restir_amd.scenes import orbit_position
    sd = get_scene("sponza:0.03")
    W, H = SCENES["sponza:0.03"]
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    for frame in range(5):
        p = orbit_position(sd.camera_args["position"], frame, radius=0.3)
        o.set_camera_position(p); h.set_camera_position(p)
        a = o.frame(3); b = h.frame(3)
        assert bits_equal(a, b), (frame, radiance_stats(a, b))
        _compare_reservoirs(o.restir.last, h.restir.download(1))


def test_restir_reset_and_accumulate(hip):
    """ReSTIRReset re-arms the first-frame flag; iter>0 takes the running mean (restir.cu:230)."""
    sd = get_scene("cornell")
    W, H = 96, 64
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    for frame in range(3):
        a = o.frame(1, iteration=frame); b = h.frame(1, iteration=frame)
        assert bits_equal(a, b)
    o.restir.reset(); h.restir.reset()
    a = o.frame(1, iteration=0); b = h.frame(1, iteration=0)
    assert bits_equal(a, b)


@pytest.mark.parametrize("name", list(SCENES))
def test_path_trace_direct(hip, name):
    """Config 1 semantics (PTDirectKernel, 1 spp, looper = 0) plus an accumulated second sample."""
    sd = get_scene(name)
    W, H = SCENES[name]
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    for it in range(2):
        a = o.frame(0, use_reservoir=False, iteration=it); b = h.frame(0, use_reservoir=False, iteration=it)
        assert o.rays == h.rays
        assert bits_equal(a, b), radiance_stats(a, b)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_copy_image_to_pbo(hip, mode):
    import torch
    rng = np.random.default_rng(7)
    W, H = 200, 100
    img = rng.uniform(0, 4, (W * H, 3)).astype(np.float32)
    img[:100] = 0; img[100:200] = 1e6; img[200:210] = np.nan; img[210:220] = -1.0
    ref = ob.send_image_to_pbo(img, W, H, mode, 0.9)
    t = torch.from_numpy(img).cuda()
    out = torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda")
    hip.copy_image_to_pbo(out.data_ptr(), t.data_ptr(), W, H, mode, 0.9)
    got = out.cpu().numpy()
    diff = np.abs(ref.astype(np.int32) - got.astype(np.int32))
    assert diff.max() <= 1
    assert np.mean(diff > 0) <= 1e-4


def test_fast_gamma_equals_exact(hip, monkeypatch):
    """copyImageToPBO decides the byte with exp2(log2(c)/2.2) and falls back to the double-precision power near the 255
    integer boundaries; the result must equal the all-exact kernel (RS_EXACT_GAMMA) on every value: dense logarithmic
    sweep over 24 decades, values straddling every byte boundary, zeros, negatives, NaN and infinities."""
    import torch
    rng = np.random.default_rng(21)
    n = 6_000_000
    x = np.exp(rng.uniform(np.log(1e-12), np.log(1e12), n)).astype(np.float32)
    b = (np.arange(1, 256, dtype=np.float64) / 255.0) ** 2.2                                 # c with c^(1/2.2) * 255 = integer
    near = (b[None, :] * (1.0 + rng.uniform(-3e-6, 3e-6, (2000, 255)))).astype(np.float32).reshape(-1)
    x[:near.size] = near
    x[near.size:near.size + 8] = [0.0, -0.0, -1.0, np.nan, np.inf, -np.inf, 1e-7, 1.0]
    img = np.ascontiguousarray(x[: (n // 3) * 3].reshape(-1, 3))
    W, H = 1000, img.shape[0] // 1000
    img = img[: W * H]
    t = torch.from_numpy(img).cuda()
    outs = []
    for exact in (False, True):
        if exact:
            monkeypatch.setenv("RS_EXACT_GAMMA", "1")
        o = torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda")
        hip.copy_image_to_pbo(o.data_ptr(), t.data_ptr(), W, H, 0, 1.0)
        outs.append(o.cpu().numpy())
    monkeypatch.delenv("RS_EXACT_GAMMA")
    assert np.array_equal(outs[0], outs[1])
    assert len(np.unique(outs[0][:, :3])) == 256


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_copy_debug_image_to_pbo(hip, kind):
    """The vec2 / float / int overloads of copyImageToPBO (pathtrace.cu:58-106), e.g. the motion-vector view."""
    import torch
    rng = np.random.default_rng(8 + kind)
    W, H = 200, 100
    if kind == 2:
        img = rng.integers(-1, W * H, W * H).astype(np.int32)                 # devMotion holds pixel indices or -1
    else:
        img = rng.uniform(-0.2, 1.5, (W * H, 2) if kind == 0 else (W * H,)).astype(np.float32)
    ref = ob.send_debug_to_pbo(img, W, H, kind)
    out = torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda")
    hip.copy_debug_image_to_pbo(out.data_ptr(), torch.from_numpy(img).cuda().data_ptr(), W, H, kind)
    diff = np.abs(ref.astype(np.int32) - out.cpu().numpy().astype(np.int32))
    assert diff.max() <= 1 and np.mean(diff > 0) <= 1e-4


@pytest.mark.parametrize("fused", [True, False], ids=["fused taps (default)", "separately rounded taps"])
def test_eaw_filter(hip, fused):
    """LeveledEAWFilter against the oracle, stated tolerance rtol 1e-5 (the exponential is the hardware's 2^t): with the taps in fused
    arithmetic (rs_eaw_set_fused, the default) and with every operation rounded separately in the reference's order -- measured
    7.4e-7 and 7.1e-7 from the oracle (tools/ab_eaw_fused.py)."""
    import torch
    sd = get_scene("sponza:0.03")
    W, H = 160, 96
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    # runCuda order: G-buffer, shading, (denoise here), gBuffer.update
    o.gbuf.render(o.scene, o.cam); h.gbuf.render(h.scene, h.cam)
    o.restir.direct(o.scene, o.cam, o.gbuf, o.image, 0, 0, 3)
    h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, 0, 3)
    a = o.image
    ref = ob.eaw_filter(o.gbuf, o.cam, a)
    f = hip.EAWFilter(W, H, 5)
    f.set_fused(fused)
    out = torch.zeros_like(h.image)
    p = f.filter(out.data_ptr(), h.image.data_ptr(), h.gbuf, h.cam)
    hip.synchronize()
    res = torch.empty_like(h.image)
    hip.hip_memcpy_d2d(res.data_ptr(), p, res.numel() * 4)
    got = res.cpu().numpy()
    f.destroy()
    assert np.allclose(ref, got, rtol=1e-5, atol=1e-6), float(np.abs(ref - got).max())
    assert float((np.abs(ref - got) / np.maximum(np.abs(ref), 1e-6)).max()) < 3e-6          # what the arithmetic actually delivers
    assert np.abs(ref - a).max() > 1e-3               # the filter did something


def test_eaw_filter_with_edited_sigmas(hip):
    """The viewer edits LeveledEAWFilter::waveletFilter.sig* between frames (src/preview.cpp:262-265): rs_eaw_set_params, with
    sigmas that are not powers of two (the kernels then divide as the reference does) and with power-of-two ones."""
    import torch
    sd = get_scene("sponza:0.03")
    W, H = 160, 96
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    o.frame(3); h.frame(3)
    f = hip.EAWFilter(W, H, 5)
    assert f.get_params() == (64.0, np.float32(0.2), 1.0, 5)
    for sig, fused in (((3.7, 0.35, 0.6), False), ((8.0, 0.25, 2.0), False), ((3.7, 0.35, 0.6), True)):
        f.set_fused(fused)
        f.set_params(*sig, level=3)                      # the level is stored, the filter still runs five (src/denoiser.cu:463-477)
        assert f.get_params()[3] == 3
        out = torch.zeros_like(h.image)
        # gbuf.update() ran in frame(): the planes the filter reads are the "last" ones now, so render again as runCuda would
        o.gbuf.render(o.scene, o.cam); h.gbuf.render(h.scene, h.cam)
        ref = ob.eaw_filter_with(o.gbuf, o.cam, o.image, *sig)
        p = f.filter(out.data_ptr(), h.image.data_ptr(), h.gbuf, h.cam)
        hip.synchronize()
        res = torch.empty_like(h.image)
        hip.hip_memcpy_d2d(res.data_ptr(), p, res.numel() * 4)
        got = res.cpu().numpy()
        assert np.allclose(ref, got, rtol=1e-5, atol=1e-6), (sig, float(np.abs(ref - got).max()))
        o.gbuf.update(o.cam); h.gbuf.update(h.cam)
    f.destroy()


@pytest.mark.parametrize("sigma", [0.2, 0.3, 3.0, 7.7, 100.0, 1e-3, 12345.678])
def test_eaw_division_by_sigma_is_exact(hip, sigma):
    """The filter divides |dn|^2 by sigNormal = 0.2 (any sigma that is not a power of two) with a reciprocal, an exact residual and
    one correction (denoiser.hip div_sigma): the IEEE quotient on every float in [2^-100, 2^100], 1.68e9 of them per sigma."""
    bad = C.c_ulonglong(1)
    hip.check(hip.lib().rs_debug_div_sigma_mismatches(C.c_float(sigma), C.byref(bad)))
    assert bad.value == 0, (sigma, bad.value)


@pytest.mark.parametrize("fused", [True, False], ids=["fused taps (default)", "separately rounded taps"])
def test_eaw_tiled_levels_equal_plain_gathers(hip, fused):
    """The levels of step 1, 2 and 4 read their taps from an LDS tile (k_wavelet_tiled); the claim is the plain kernel's arithmetic in
    the same order, so the filtered image must be the plain form's BIT FOR BIT -- checked directly, not through the oracle's rtol:
    the default sigmas and two edited sets (other division / multiplication instantiations), a frame size that is no multiple of
    the 32x8 block, and the row-strip form on a row range that starts in the middle of a block."""
    import torch
    sd = get_scene("sponza:0.03")
    W, H = 157, 83
    h = HipRenderer(hip, sd, W, H)
    h.frame(3); h.frame(3)
    h.gbuf.render(h.scene, h.cam)                        # the planes the filter reads are this frame's
    f = hip.EAWFilter(W, H, 5)
    f.set_fused(fused)

    def filtered(tiled):
        f.set_tiled(tiled)
        out = torch.zeros_like(h.image)
        p = f.filter(out.data_ptr(), h.image.data_ptr(), h.gbuf, h.cam)
        hip.synchronize()
        res = torch.empty_like(h.image)
        hip.hip_memcpy_d2d(res.data_ptr(), p, res.numel() * 4)
        torch.cuda.synchronize()
        return res.cpu().numpy()

    for sig in (None, (3.7, 0.35, 0.6), (8.0, 0.25, 2.0)):
        if sig is not None:
            f.set_params(*sig, level=5)
        a, b = filtered(True), filtered(False)
        assert np.isfinite(a).all() and np.abs(a - h.image.cpu().numpy()).max() > 1e-4, sig      # the filter did something
        assert bits_equal(a, b), (sig, int((a.view(np.uint32) != b.view(np.uint32)).sum()))
    # row-strip form, rows [13, 70): level by level into full-frame buffers, tiled against plain
    y0, y1 = 13, 70
    outs = []
    for tiled in (True, False):
        f.set_tiled(tiled)
        f.positions_rows(h.gbuf, h.cam, 0, H)
        bufs = [torch.zeros_like(h.image), torch.zeros_like(h.image)]
        src = h.image
        for level in range(5):
            # every level reads rows outside [y0, y1): give it a full-frame input (level 0: the image; later: the previous output
            # over the whole frame, computed with the same form)
            f.level_rows(bufs[level % 2].data_ptr(), src.data_ptr(), h.gbuf, level, 0, H)
            src = bufs[level % 2]
        full = src.clone()
        part = torch.zeros_like(h.image)
        # the last level once more on the row range only, from the same input: must equal those rows of the full-frame pass
        last_in = bufs[(4 - 1) % 2]
        f.level_rows(part.data_ptr(), last_in.data_ptr(), h.gbuf, 4, y0, y1)
        hip.synchronize(); torch.cuda.synchronize()
        assert bits_equal(part[y0 * W:y1 * W].cpu().numpy(), full[y0 * W:y1 * W].cpu().numpy())
        outs.append(full.cpu().numpy())
    assert bits_equal(outs[0], outs[1])
    # a low level on a range that starts inside a block row, tiled against plain
    rows = []
    for tiled in (True, False):
        f.set_tiled(tiled)
        o = torch.zeros_like(h.image)
        f.level_rows(o.data_ptr(), h.image.data_ptr(), h.gbuf, 1, y0, y1)
        hip.synchronize(); torch.cuda.synchronize()
        rows.append(o[y0 * W:y1 * W].cpu().numpy())
    assert bits_equal(rows[0], rows[1])
    f.destroy()


def test_fused_taps_stay_within_ulps_of_the_separately_rounded_ones(hip):
    """rs_eaw_set_fused / rs_svgf_set_fused change roundings, not the filter: on the same input the two forms of LeveledEAWFilter agree
    to 5e-6 relative after five levels, those of SpatioTemporalFilter to 2e-5 after five frames of accumulated history."""
    import torch
    sd = get_scene("sponza:0.03")
    W, H = 157, 83
    h = HipRenderer(hip, sd, W, H)
    h.frame(3); h.frame(3)
    h.gbuf.render(h.scene, h.cam)

    def grab(ptr, count):
        t = torch.empty(count, dtype=torch.float32, device="cuda")
        hip.hip_memcpy_d2d(t.data_ptr(), ptr, count * 4)
        return t.cpu().numpy()

    outs = []
    for fused in (True, False):
        f = hip.EAWFilter(W, H, 5)
        f.set_fused(fused)
        out = torch.zeros_like(h.image)
        outs.append(grab(f.filter(out.data_ptr(), h.image.data_ptr(), h.gbuf, h.cam), W * H * 3))
        f.destroy()
    assert not bits_equal(outs[0], outs[1])                   # the switch acts
    assert float((np.abs(outs[0] - outs[1]) / np.maximum(np.abs(outs[1]), 1e-6)).max()) < 5e-6
    filters = [hip.SVGFFilter(W, H, 5), hip.SVGFFilter(W, H, 5)]
    filters[0].set_fused(True); filters[1].set_fused(False)
    for frame in range(5):
        res = []
        for f in filters:
            res.append(grab(f.filter(h.image.data_ptr(), h.gbuf, h.cam), W * H * 3))
            f.next_frame()
        assert float((np.abs(res[0] - res[1]) / np.maximum(np.abs(res[1]), 1e-6)).max()) < 2e-5, frame
    assert not bits_equal(res[0], res[1])
    for f in filters:
        f.destroy()


def test_eaw_row_strip_form_equals_filter(hip):
    """rs_eaw_positions_rows + rs_eaw_level_rows (the form the strip tiling drives, restir_amd/tiling.py eaw_filter) over the
    whole frame, in bands, give the image of rs_eaw_filter bit for bit."""
    import torch
    from restir_amd.tiling import HipBackend, StripRenderer
    sd = get_scene("sponza:0.03")
    W, H = 160, 96
    scene = hip_scene(hip, sd)
    cam = hip.camera_update(sd.camera(W, H))
    b = HipBackend(hip, scene, cam, W, H)
    s = StripRenderer(b, 1, 0, H)
    b.gbuffer_render(0, H); b.phase_a(0, 3, 0, H); b.phase_b(0, 3, 0, H)
    f = hip.EAWFilter(W, H, 5)
    out = torch.zeros_like(b.image)
    p = f.filter(out.data_ptr(), b.image.data_ptr(), b.gbuf, cam)
    ref = torch.empty_like(b.image); hip.hip_memcpy_d2d(ref.data_ptr(), p, ref.numel() * 4)
    got = s.eaw_filter().clone()
    assert bits_equal(ref.cpu().numpy(), got.cpu().numpy())
    # the same levels in three row bands
    b.eaw_positions(0, H)
    for level in range(5):
        for y0, y1 in ((40, 77), (0, 40), (77, H)):
            b.eaw_level(level, y0, y1)
    assert bits_equal(ref.cpu().numpy(), b.eaw_result().cpu().numpy())
    f.destroy()


@pytest.mark.parametrize("fused", [True, False], ids=["fused taps (default)", "separately rounded taps"])
def test_svgf_filter(hip, fused):
    """SpatioTemporalFilter (denoiser.cu:136-216,250-371,479-568) on an orbiting camera: temporal accumulation through
    devMotion, spatial then (from the fifth frame on) temporal variance, five variance-guided a-trous levels, and
    the pointer hand-over of filter(); filtered image and filter state against the oracle every frame."""
    import torch
    from restir_amd.scenes import orbit_position
    sd = get_scene("sponza:0.03")
    W, H = 160, 96
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    fo = ob.SVGF(W, H)
    fh = hip.SVGFFilter(W, H, 5)
    fh.set_fused(fused)

    def grab(ptr, count):
        t = torch.empty(count, dtype=torch.float32, device="cuda")
        hip.hip_memcpy_d2d(t.data_ptr(), ptr, count * 4)
        return t.cpu().numpy()

    for frame in range(7):
        p = orbit_position(sd.camera_args["position"], frame, radius=0.3)
        o.set_camera_position(p); h.set_camera_position(p)
        # runCuda order: G-buffer, shading, (denoise here), gBuffer.update
        o.gbuf.render(o.scene, o.cam); h.gbuf.render(h.scene, h.cam)
        o.restir.direct(o.scene, o.cam, o.gbuf, o.image, 0, o.looper, 1)
        h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, h.looper, 1)
        o.looper += 1; h.looper += 1
        ref = fo.filter(o.image, o.gbuf, o.cam)
        got = grab(fh.filter(h.image.data_ptr(), h.gbuf, h.cam), W * H * 3).reshape(-1, 3)
        assert np.allclose(ref, got, rtol=3e-5, atol=2e-6), (frame, float(np.abs(ref - got).max()))
        st = fo.state(); v = fh.view()
        assert v.frameIdx == frame % 2
        assert np.allclose(st["variance"], grab(v.devVariance, W * H), rtol=3e-5, atol=1e-6), frame
        assert np.allclose(st["accum_moment"], grab(v.devAccumMoment[v.frameIdx], W * H * 3).reshape(-1, 3), rtol=1e-6, atol=1e-7), frame
        assert np.allclose(st["accum_color"], grab(v.devAccumColor[v.frameIdx], W * H * 3).reshape(-1, 3), rtol=3e-5, atol=2e-6), frame
        if frame >= 5:
            assert (st["accum_moment"][:, 2] > 3.5).mean() > 0.3          # the temporal-variance branch is exercised
        fo.next_frame(); fh.next_frame()
        o.gbuf.update(o.cam); h.gbuf.update(h.cam)
    assert np.abs(ref - o.image).max() > 1e-3                              # the filter did something
    fh.destroy()


@pytest.mark.parametrize("fused", [True, False], ids=["fused taps (default)", "separately rounded taps"])
def test_svgf_tiled_levels_equal_plain_gathers(hip, fused):
    """SpatioTemporalFilter's variance-guided a-trous levels from the row-phase LDS tile (k_svgf_wavelet_tiled) against the plain
    gathers: the same arithmetic in the same order, so filtered image, variance and history must agree BIT FOR BIT over several
    frames of a moving camera, on a frame size that is no multiple of the tile."""
    import torch
    from restir_amd.scenes import orbit_position
    sd = get_scene("sponza:0.03")
    W, H = 157, 83
    h = HipRenderer(hip, sd, W, H)
    filters = [hip.SVGFFilter(W, H, 5), hip.SVGFFilter(W, H, 5)]
    filters[1].set_tiled(False)
    for f in filters:
        f.set_fused(fused)

    def grab(ptr, count):
        t = torch.empty(count, dtype=torch.float32, device="cuda")
        hip.hip_memcpy_d2d(t.data_ptr(), ptr, count * 4)
        return t.cpu().numpy()

    for frame in range(6):
        h.set_camera_position(orbit_position(sd.camera_args["position"], frame, radius=0.3))
        h.gbuf.render(h.scene, h.cam)
        h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, h.looper, 1)
        h.looper += 1
        outs = []
        for f in filters:
            img = grab(f.filter(h.image.data_ptr(), h.gbuf, h.cam), W * H * 3)
            v = f.view()
            outs.append((img, grab(v.devVariance, W * H), grab(v.devAccumColor[v.frameIdx], W * H * 3)))
            f.next_frame()
        for a, b in zip(outs[0], outs[1]):
            assert bits_equal(a, b), (frame, int((a.view(np.uint32) != b.view(np.uint32)).sum()))
        assert np.isfinite(outs[0][0]).all() and np.abs(outs[0][0] - h.image.cpu().numpy().reshape(-1)).max() > 1e-4
        h.gbuf.update(h.cam)


def test_modulate_and_add(hip):
    import torch
    sd = get_scene("cornell")
    W, H = 64, 64
    o = OracleRenderer(sd, W, H); h = HipRenderer(hip, sd, W, H)
    o.gbuf.render(o.scene, o.cam); h.gbuf.render(h.scene, h.cam)
    rng = np.random.default_rng(1)
    img = rng.uniform(0, 0.95, (W * H, 3)).astype(np.float32)
    other = rng.uniform(0, 1, (W * H, 3)).astype(np.float32)
    ref = img.copy(); ob.lib().orc_modulate(W, H, ref.reshape(-1), o.gbuf.albedo.reshape(-1))
    t = torch.from_numpy(img).cuda()
    hip.check(hip.lib().rs_modulate_albedo(t.data_ptr(), h.gbuf.handle))
    assert bits_equal(ref, t.cpu().numpy())
    t2 = torch.from_numpy(other).cuda()
    hip.check(hip.lib().rs_add_image(t.data_ptr(), t2.data_ptr(), W, H))
    assert bits_equal(ref + other, t.cpu().numpy())
    t3 = torch.empty_like(t)
    hip.check(hip.lib().rs_add_image3(t3.data_ptr(), t.data_ptr(), t2.data_ptr(), W, H))
    assert bits_equal((ref + other) + other, t3.cpu().numpy())


def test_strip_tiling_equals_full_frame(hip):
    """Framebuffer row strips with a 5-row reservoir halo (the multi-GPU decomposition) reproduce the
    full-frame result bit for bit.  Two 'ranks' are emulated on one GPU, each with its own buffers."""
    import torch
    sd = get_scene("sponza:0.03")
    W, H = 160, 96
    full = HipRenderer(hip, sd, W, H)
    ranks = [HipRenderer(hip, sd, W, H, scene=full.scene) for _ in range(2)]
    bounds = [(0, 40), (40, H)]
    halo = 5
    for frame in range(3):
        ref = full.frame(3)
        # phase A on each strip
        for r, (y0, y1) in zip(ranks, bounds):
            r.gbuf.render(r.scene, r.cam, y0, y1)                                   # its own rows only
            r.restir.phase_a(r.scene, r.cam, r.gbuf, r.looper, 3, y0, y1)
        # halo exchange (what RCCL send/recv carries between neighbours): published reservoirs + G-buffer id / normal / depth rows
        nr, ng = ranks[0].restir.halo_bytes(halo), ranks[0].gbuf.rows_bytes(halo)
        up = torch.empty(nr + ng, dtype=torch.uint8, device="cuda"); down = torch.empty(nr + ng, dtype=torch.uint8, device="cuda")
        ranks[0].restir.halo_pack(bounds[0][1] - halo, halo, down.data_ptr())       # rank0's last rows -> rank1
        ranks[0].gbuf.rows_pack(0, bounds[0][1] - halo, halo, down.data_ptr() + nr)
        ranks[1].restir.halo_pack(bounds[1][0], halo, up.data_ptr())                # rank1's first rows -> rank0
        ranks[1].gbuf.rows_pack(0, bounds[1][0], halo, up.data_ptr() + nr)
        ranks[1].restir.halo_unpack(bounds[0][1] - halo, halo, down.data_ptr())
        ranks[1].gbuf.rows_unpack(0, bounds[0][1] - halo, halo, down.data_ptr() + nr)
        ranks[0].restir.halo_unpack(bounds[1][0], halo, up.data_ptr())
        ranks[0].gbuf.rows_unpack(0, bounds[1][0], halo, up.data_ptr() + nr)
        for r, (y0, y1) in zip(ranks, bounds):
            r.restir.phase_b(r.scene, r.cam, r.gbuf, r.image.data_ptr(), 0, 3, y0, y1)
            r.restir.end_frame()
            r.looper += 1
            r.gbuf.update(r.cam)
        hip.synchronize()
        got = np.concatenate([ranks[0].image.cpu().numpy()[:bounds[0][1] * W], ranks[1].image.cpu().numpy()[bounds[1][0] * W:]])
        assert bits_equal(ref, got), (frame, radiance_stats(ref, got))


def test_internal_streams_are_chosen_by_measurement_and_change_no_result(hip):
    """The library chooses its three internal streams by measurement when an overlapped frame first needs them, again after
    rs_choose_internal_streams_again / rs_set_internal_stream_priority / rs_set_stream with another stream (api_common.hip).  Which streams
    carry the chains is scheduling only: overlapped frames under every preference, on the default stream and on an ordinary one, with the
    choice repeated in the middle of the sequence, equal the synchronous frames bit for bit; rs_internal_streams_info reports a level and a
    calibration time."""
    import torch
    from restir_amd.scenes import orbit_position
    sd = get_scene("sponza:0.1")
    W, H, frames = 320, 180, 9
    scene = hip_scene(hip, sd)

    def run(overlapped, level=2, own_stream=False):
        stream = torch.cuda.Stream() if own_stream else torch.cuda.current_stream()
        with torch.cuda.stream(stream):
            hip.set_stream(stream.cuda_stream)                     # the library and torch's copies of the images on ONE stream
            h = HipRenderer(hip, sd, W, H, scene=scene)
            images = []
            hip.set_sync(not overlapped)
            hip.set_internal_stream_priority(level)
            try:
                for frame in range(frames):
                    if overlapped and frame == 4:
                        hip.choose_internal_streams_again()
                    h.set_camera_position(orbit_position(sd.camera_args["position"], frame, radius=0.5))
                    h.gbuf.render(h.scene, h.cam)
                    h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, h.looper, 3)
                    h.looper += 1
                    images.append(h.image.clone())
                    h.gbuf.update(h.cam)
                hip.synchronize()
                torch.cuda.synchronize()
                info = hip.internal_streams_info()
            finally:
                hip.set_sync(True)
                hip.set_internal_stream_priority(2)
        hip.set_stream(torch.cuda.current_stream().cuda_stream)
        return [t.cpu().numpy() for t in images], h.restir.download(1), info

    ref_images, ref_resv, _ = run(False)
    for level, own in ((2, False), (1, False), (0, True), (-1, True), (1, True)):
        images, resv, info = run(True, level, own)
        for f in range(frames):
            assert bits_equal(images[f], ref_images[f]), (level, own, f)
        for k in resv.dtype.names:
            assert bits_equal(resv[k], ref_resv[k]), (level, own, k)
        lvl, chosen_us, fastest_us = info
        assert lvl in (-1, 0, 1)
        assert chosen_us == 0.0 or (0.0 < fastest_us <= chosen_us <= fastest_us * 1.08 + 1e-6), info     # 0: plain streams (nothing could be measured)


def test_spatial_pass_event_ring_and_probe_name(hip):
    """Measurement hooks of bench.py: rs_restir_enable_timing(r, 2) keeps the two events around the spatial pass of every frame in a ring that
    is read afterwards (nothing waits inside the frames), and rs_restir_set_probe(r, 1) launches the pass under another kernel name -- the same
    code: frames rendered either way are bit-identical."""
    import torch
    sd = get_scene("sponza:0.1")
    W, H = 320, 180
    scene = hip_scene(hip, sd)

    def run(probe, timing):
        h = HipRenderer(hip, sd, W, H, scene=scene)
        hip.set_sync(False)
        try:
            h.restir.set_probe(probe)
            if timing:
                h.restir.enable_timing(2)
            for frame in range(7):
                h.gbuf.render(h.scene, h.cam)
                h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, frame, 3)
                h.gbuf.update(h.cam)
            times = h.restir.spatial_times(64) if timing else []
            hip.synchronize(); torch.cuda.synchronize()
        finally:
            h.restir.set_probe(False); h.restir.enable_timing(False)
            hip.set_sync(True)
        return h.image.cpu().numpy(), h.restir.download(1), times

    a = run(False, False)
    b = run(True, True)
    assert bits_equal(a[0], b[0]) and a[1].tobytes() == b[1].tobytes()
    assert len(b[2]) == 7 and all(0.0 < t < 50.0 for t in b[2]), b[2]


def test_stream_choice_is_kept_per_caller_stream(hip):
    """rs_prepare_streams makes the choice at once; rs_set_stream back to a stream the library has measured next to takes that choice again
    instead of measuring (about 25 ms) once more -- a caller that alternates between two streams pays twice, not at every switch."""
    import time
    import torch
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    hip.set_sync(False)
    try:
        took = []
        for st in (a, b, a, b, a):
            hip.set_stream(st.cuda_stream)
            t0 = time.perf_counter()
            hip.prepare_streams()
            took.append((time.perf_counter() - t0) * 1e3)
            lvl, chosen_us, fastest_us = hip.internal_streams_info()
            assert lvl in (-1, 0, 1)
        measured = took[:2]
        if min(measured) > 5.0:                                    # (the measurement ran: a box where it fails uses plain streams and has nothing to keep)
            assert max(took[2:]) < 0.5 * min(measured), took
    finally:
        hip.set_sync(True)
        hip.set_stream(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("fused", [False, True])
def test_overlapped_frames_equal_synchronous_frames(hip, fused):
    """Asynchronous mode (rs_set_sync(0)) lets frames overlap: GBuffer::render and the primary-ray + RIS kernels of frame f + 1
    run on auxiliary streams next to the temporal / spatial passes of frame f (G-buffer planes in a ring of three, surface
    planes double-buffered, events for every true dependency).  Twelve frames of an orbiting camera enqueued without any host
    synchronisation, with the EAW filter and the tone map reading each frame's planes, must equal the synchronous run bit
    for bit -- images, filtered images, final reservoirs and the G-buffer."""
    import torch
    from restir_amd.scenes import orbit_position
    sd = get_scene("sponza:0.2")
    W, H, frames = 640, 360, 12
    scene = hip_scene(hip, sd)

    def run(overlapped):
        h = HipRenderer(hip, sd, W, H, scene=scene)
        f = hip.EAWFilter(W, H, 5)
        out = torch.zeros_like(h.image)
        images, filtered, pbos = [], [], []
        hip.set_sync(not overlapped)
        hip.set_side_stream(3 if fused else 1)                   # 3: GBuffer::render deferred into the primary-ray launch (one walk for both rays), any size
        try:
            for frame in range(frames):
                h.set_camera_position(orbit_position(sd.camera_args["position"], frame, radius=0.5))
                h.gbuf.render(h.scene, h.cam)
                if frame % 5 == 3:                                       # a second render without an update in between
                    h.gbuf.render(h.scene, h.cam, 0, H // 2)
                h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, h.looper, 3)
                h.looper += 1
                images.append(h.image.clone())                           # enqueued on the library (= torch's current) stream
                p = f.filter(out.data_ptr(), h.image.data_ptr(), h.gbuf, h.cam)
                t = torch.empty_like(h.image)
                hip.hip_memcpy_d2d_async(t.data_ptr(), p, t.numel() * 4)
                filtered.append(t)
                pbo = torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda")
                hip.copy_image_to_pbo(pbo.data_ptr(), h.image.data_ptr(), W, H, 2, 1.0)
                pbos.append(pbo)
                h.gbuf.update(h.cam)
            hip.synchronize()
            torch.cuda.synchronize()
        finally:
            hip.set_sync(True)
            hip.set_side_stream(4)
        res = dict(images=[t.cpu().numpy() for t in images], filtered=[t.cpu().numpy() for t in filtered],
                   pbos=[t.cpu().numpy() for t in pbos], resv=h.restir.download(1), gbuf=h.gbuf.download())
        f.destroy()
        return res

    a, b = run(False), run(True)
    for k in ("images", "filtered", "pbos"):
        for frame in range(frames):
            assert bits_equal(a[k][frame], b[k][frame]), (k, frame)
    assert a["resv"].tobytes() == b["resv"].tobytes()
    for k in ("albedo", "motion"):
        assert bits_equal(a["gbuf"][k], b["gbuf"][k]), k
    for k in ("normal", "prim_id", "depth"):
        for i in range(2):
            assert bits_equal(a["gbuf"][k][i], b["gbuf"][k][i]), (k, i)
    assert not bits_equal(a["images"][0], a["images"][-1])


@pytest.mark.parametrize("join_every_frame", [False, True])
def test_denoise_stream_equals_synchronous_frames(hip, join_every_frame):
    """rs_set_denoise_stream(1): LeveledEAWFilter of frame f and the tone map of its result run on a stream of the library next to the
    passes of frame f + 1 (src/main.cpp:160-181's order of calls, unchanged for the caller).  Fourteen frames of an orbiting camera
    enqueued without a host synchronisation -- and, in the first variant, without the library stream ever waiting for the denoise
    stream, so that a phase B that overwrote the image an earlier filter still reads, or a render that overwrote its G-buffer set,
    would show -- give the display image of EVERY frame, the last filtered image, the radiance and the reservoirs of the synchronous
    run, bit for bit.  Second variant: the caller's own copy of every filtered image after rs_join_denoise_stream()."""
    import torch
    from restir_amd.scenes import orbit_position
    sd = get_scene("sponza:0.2")
    W, H, frames = 640, 360, 14
    scene = hip_scene(hip, sd)

    def run(overlapped):
        h = HipRenderer(hip, sd, W, H, scene=scene)
        f = hip.EAWFilter(W, H, 5)
        out = torch.zeros_like(h.image)
        pbos = [torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda") for _ in range(frames)]
        filtered = []
        hip.set_sync(not overlapped)
        hip.set_denoise_stream(1 if overlapped else 0)
        try:
            for frame in range(frames):
                h.set_camera_position(orbit_position(sd.camera_args["position"], frame, radius=0.5))
                h.gbuf.render(h.scene, h.cam)
                h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, h.looper, 3)
                h.looper += 1
                p = f.filter(out.data_ptr(), h.image.data_ptr(), h.gbuf, h.cam)
                hip.copy_image_to_pbo(pbos[frame].data_ptr(), p, W, H, 2, 1.0)
                if join_every_frame:
                    t = torch.empty_like(h.image)
                    if overlapped:
                        hip.join_denoise_stream()
                    hip.hip_memcpy_d2d_async(t.data_ptr(), p, t.numel() * 4)          # on the library (= torch's current) stream
                    filtered.append(t)
                h.gbuf.update(h.cam)
            hip.synchronize()
            torch.cuda.synchronize()
            last = torch.empty_like(h.image)
            hip.hip_memcpy_d2d(last.data_ptr(), p, last.numel() * 4)
            form = h.restir.last_launch()
        finally:
            hip.set_sync(True)
            hip.set_denoise_stream(0)
        res = dict(image=h.image.cpu().numpy(), last=last.cpu().numpy(), pbos=[t.cpu().numpy() for t in pbos],
                   filtered=[t.cpu().numpy() for t in filtered], resv=h.restir.download(1), form=form)
        f.destroy()
        return res

    a, b = run(False), run(True)
    assert b["form"][1] <= 2, b["form"]                               # the chains of the frames take turns on two streams next to the denoise stream
    assert bits_equal(a["image"], b["image"]) and bits_equal(a["last"], b["last"])
    for frame in range(frames):
        assert bits_equal(a["pbos"][frame], b["pbos"][frame]), frame
    for frame in range(len(a["filtered"])):
        assert bits_equal(a["filtered"][frame], b["filtered"][frame]), frame
    assert a["resv"].tobytes() == b["resv"].tobytes()
    assert not bits_equal(a["pbos"][0], a["pbos"][-1]) and a["pbos"][-1].any()


def test_measured_choice_of_the_fused_walk_keeps_the_images(hip):
    """Default asynchronous mode: ReSTIRDirect measures once per scene whether walking the G-buffer ray with the shading ray is
    faster (frames 18..33 run fused, all others until the decision separately).  Whatever it picks, a full-size run of
    40 frames equals the synchronous run bit for bit."""
    import torch
    from restir_amd.scenes import orbit_position
    sd = get_scene("sponza:0.1")
    W, H, frames = 1920, 1080, 40
    scene = hip_scene(hip, sd)

    def run(overlapped):
        h = HipRenderer(hip, sd, W, H, scene=scene)
        keep = []
        hip.set_sync(not overlapped)
        hip.set_side_stream(4)
        try:
            for frame in range(frames):
                h.set_camera_position(orbit_position(sd.camera_args["position"], frame, radius=0.3))
                h.gbuf.render(h.scene, h.cam)
                h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, h.looper, 3)
                h.looper += 1
                if frame in (3, 11, 17, 19, 27, 33, 35, 39):
                    keep.append(h.image.clone())
                h.gbuf.update(h.cam)
            hip.synchronize(); torch.cuda.synchronize()
            out = [t.cpu().numpy() for t in keep] + [h.restir.download(1).view(np.uint8), h.gbuf.download()["depth"][0]]
            # the library never waits on the host for its measurement: it asks at every frame end whether the last time stamp has
            # been reached, so the decision falls at the first frame end after the GPU has caught up with frame 34
            h.gbuf.render(h.scene, h.cam)
            h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, h.looper, 3)
            h.gbuf.update(h.cam)
            hip.synchronize()
        finally:
            hip.set_sync(True)
        choices.append(h.restir.launch_choice())
        return out

    choices = []
    for a, b in zip(run(False), run(True)):
        assert bits_equal(a, b)
    assert choices[0] == -2 and choices[1] in (0, 1)          # synchronous launches: nothing to choose; overlapped: decided once the stamp of frame 34 is there


def test_scene_destroyed_while_a_render_is_only_recorded(hip):
    """Asynchronous mode defers GBuffer::render (it is launched by the next reader of the planes).  Destroying the scene in
    between must launch the recorded render while the scene still exists: the planes then equal those of a synchronous
    render, and nothing dereferences the freed scene afterwards (update, a second scene, more renders)."""
    sd = get_scene("cornell")
    W, H = 96, 64
    ref = HipRenderer(hip, sd, W, H)
    ref.gbuf.render(ref.scene, ref.cam)
    want = ref.gbuf.download()
    scene = hip_scene(hip, sd)
    h = HipRenderer(hip, sd, W, H, scene=scene)
    hip.set_sync(False)
    hip.set_side_stream(3)
    try:
        h.gbuf.render(scene, h.cam)                 # recorded only
        scene.destroy()                              # launches it, waits, frees the arrays
        h.scene = None
        got = h.gbuf.download()
        for k in ("albedo", "motion"):
            assert bits_equal(got[k], want[k]), k
        for k in ("normal", "prim_id", "depth"):
            assert bits_equal(got[k][0], want[k][0]), k
        h.gbuf.update(h.cam)
        other = hip_scene(hip, sd)                   # may reuse the freed scene's address
        h.gbuf.render(other, h.cam)
        h.restir.direct(other, h.cam, h.gbuf, h.image.data_ptr(), 0, 0, 3)
        h.gbuf.update(h.cam)
        hip.synchronize()
    finally:
        hip.set_sync(True)
        hip.set_side_stream(4)


def test_phase_b_in_row_bands_equals_one_call(hip):
    """The overlapped multi-GPU schedule runs phase B as interior rows + two 5-row border bands (tiling.py); any split
    of the rows into bands must give the image of a single call."""
    from restir_amd.tiling import HipBackend
    sd = get_scene("sponza:0.03")
    W, H = 160, 96
    scene = hip_scene(hip, sd)
    cam = hip.camera_update(sd.camera(W, H))
    one, bands = HipBackend(hip, scene, cam, W, H), HipBackend(hip, scene, cam, W, H)
    for frame in range(3):
        for b in (one, bands):
            b.gbuffer_render(0, H); b.phase_a(frame, 3, 0, H)
        one.phase_b(frame, 3, 0, H)                      # iter = frame: the accumulating form (restir.cu:230)
        for y0, y1 in ((5, 91), (0, 5), (91, 96)) if frame != 1 else ((37, 38), (0, 37), (38, 96)):
            bands.phase_b(frame, 3, y0, y1)
        for b in (one, bands):
            b.end_frame()
        hip.synchronize()
        assert bits_equal(one.image.cpu().numpy(), bands.image.cpu().numpy()), frame


def test_multi_process_strips_on_one_gpu():
    """The real multi-process path (one process per rank, StripRenderer + HipBackend + torch.distributed point-to-point
    and all-gather) with two ranks sharing this GPU and gloo standing in for RCCL: tools/rehearse_strips.py compares
    the gathered strips with a full-frame render, static and orbiting camera, synchronous launches and overlapped frames,
    radiance and the EAW-filtered image (BASELINE config 5's denoiser on strips); then the same strips through the C-ABI strip
    driver (include/restir_hip.h rs_strips_frame / _eaw_filter / _exchange_history / _gather / _gather_begin,_end), the form a C++
    caller runs over RCCL; and SpatioTemporalFilter on the strips (rs_strips_svgf_filter / _exchange_svgf_history) against the
    full-frame filter, five frames of a static and of an orbiting camera."""
    import socket, subprocess, sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tools", "rehearse_strips.py")],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("strips == full frame: True") == 10, r.stdout[-2000:]
    # ... of which four lines are the C-ABI strip driver (rs_comm / rs_strips with gloo under its transport callbacks: static and
    # orbiting camera, the EAW filter, the image assembled by rs_strips_gather), also compared with tiling.py's strips on every rank
    assert r.stdout.count("C-ABI strip driver") == 7 and r.stdout.count("== tiling.py on every rank: True") == 4, r.stdout[-2000:]
    # ... and three SpatioTemporalFilter on the strips (static / orbiting camera, synchronous / overlapped launches)
    assert r.stdout.count("gathered strips == full-frame filter over 5 frames: True") == 3, r.stdout[-2000:]


def test_scene_without_extent_renders_without_the_shadow_tree(hip):
    """A scene whose vertices all coincide has no extent to lay the shadow tree's 16-bit grid over.  The reference renders it
    (its walk needs no grid: every ray misses the degenerate triangles), so scene creation must succeed with the fast path off
    (it used to fail with 'degenerate scene bounds') and frames must equal the oracle's."""
    from restir_amd.scenes import SceneData, TriangleSoup, make_materials, LAMBERTIAN, LIGHT
    soup = TriangleSoup()
    p = np.array([0.25, 0.5, -2.0], np.float32)
    for k in range(4):
        soup.add(np.array([[p, p, p]]), np.array([[[0, 0, 1]] * 3], np.float32), 0 if k < 3 else 1)
    sd = SceneData("one_point", soup, make_materials([dict(type=LAMBERTIAN, baseColor=(0.7, 0.7, 0.7)), dict(type=LIGHT, baseColor=(5.0, 5.0, 5.0))]),
                   dict(position=(0.0, 0.5, 1.0), rotation=(-90.0, 0.0, 0.0), fov_y=30.0, focal_dist=1.0))
    W, H = 64, 48
    o = OracleRenderer(sd, W, H)
    h = HipRenderer(hip, sd, W, H)
    for frame in range(2):
        a, b = o.frame(3), h.frame(3)
        assert bits_equal(a, b) and o.rays == h.rays
    seg = np.array([[0, 0, 0, 1, 1, 1], [0.25, 0.5, 0.0, 0.25, 0.5, -4.0]], np.float32)
    import torch
    assert np.array_equal(hip.trace_occlusion(h.scene, torch.from_numpy(seg).cuda()).cpu().numpy(), o.scene.test_occlusion(seg))


def test_two_contexts_in_two_threads(hip):
    """Device, stream, launch mode and the internal streams are per context, not per process: two host threads, each with its own
    context -- one synchronous like the reference, one with overlapped frames -- render the same orbit at the same time with their
    own objects, and both get the frames of a single-threaded run on the default context."""
    import threading
    import torch
    from restir_amd.scenes import orbit_position
    sd = get_scene("sponza:0.05")
    W, H, frames = 320, 180, 10

    def orbit(h):
        out = []
        for frame in range(frames):
            h.set_camera_position(orbit_position(sd.camera_args["position"], frame, radius=0.4))
            h.gbuf.render(h.scene, h.cam)
            h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, frame, 3)
            h.gbuf.update(h.cam)
            if frame % 3 == 0:
                out.append(h.image.clone())
        hip.synchronize(); torch.cuda.synchronize()
        return [t.cpu().numpy() for t in out]

    ref = orbit(HipRenderer(hip, sd, W, H))
    results, errors = {}, []

    def worker(name, overlapped):
        try:
            torch.cuda.set_device(0)
            ctx = hip.Context(0)
            ctx.make_current()
            hip.set_sync(not overlapped)
            h = HipRenderer(hip, sd, W, H)             # scene, G-buffer, reservoirs of this context
            results[name] = orbit(h)
            del h
            hip.Context.use_default()
            ctx.destroy()
        except Exception as e:                          # pragma: no cover
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=worker, args=("synchronous", False)), threading.Thread(target=worker, args=("overlapped", True))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for name in ("synchronous", "overlapped"):
        for a, b in zip(ref, results[name]):
            assert bits_equal(a, b), name
    # the default context kept its own mode: synchronous launches
    h = HipRenderer(hip, sd, 96, 64)
    h.frame(3)
    assert h.restir.launch_choice() == -2


def test_strip_driver_binds_to_rccl():
    """The C++ caller of the strip driver over RCCL in the form one GPU can run (restir_amd/host/strips_rccl_check.cpp): a
    one-rank ncclComm, 1 MiB sent to the rank itself through the library's run-time binding to librccl (ncclSend / ncclRecv in a
    group, ordered by events as in a frame), and a strip frame of a one-rank world equal to rs_restir_direct."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "restir_amd", "host", "strips_rccl_check")
    if not os.path.exists(exe):
        pytest.skip("restir_amd/host/strips_rccl_check is built only where RCCL's development files are installed (restir_amd/csrc/Makefile)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "strips_rccl_check ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_strip_driver_ranks_as_threads_over_rccl():
    """restir_amd/host/strips_rccl_ranks.cpp: N ranks as N host threads, thread k on GPU k with its own library context and its own
    ncclComm_t, the per-frame calls of INTEGRATION.md, rank 0 comparing the gathered strips with its own full frame bit for bit (static and
    orbiting camera, asynchronous launches).  Here N = 1 -- what one GPU can run (RCCL refuses two ranks on one device); on a multi-GPU node
    the same binary without an argument uses every GPU."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "restir_amd", "host", "strips_rccl_ranks")
    if not os.path.exists(exe):
        pytest.skip("restir_amd/host/strips_rccl_ranks is built only where RCCL's development files are installed (restir_amd/csrc/Makefile)")
    r = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "strips_rccl_ranks ok (1 rank)" in r.stdout and r.stdout.count("== full frame over 4 frames: True") == 2, r.stdout[-2000:] + r.stderr[-2000:]


def test_strip_driver_stream_ordered_ranks_on_one_gpu():
    """restir_amd/host/strips_loopback_ranks.cpp: three ranks as three host threads on THIS GPU, each with its own library context and
    stream, over an in-process transport with RCCL's contract (send / recv only enqueued on the given stream, grouped, matched in order per
    pair) that really moves the data -- the stream-ordered path of the strip driver, which RCCL itself can only run with one rank per
    device: border rows, the display gather DEFERRED into the next frame's group (transfers on the library stream) or on the driver's own
    stream, history exchange, EAW level rows.  bench.py's per-frame calls; every frame's gathered radiance / filtered image and every
    frame's asynchronously gathered display image equal rank 0's own full frame, static and orbiting camera: 8 modes x 6 frames; then 4 modes with the filter, the tone map of its
    result and the display gather on the library's denoise stream (rs_set_denoise_stream(1)), frames enqueued without the library stream ever
    waiting for that stream."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "restir_amd", "host", "strips_loopback_ranks")
    assert os.path.exists(exe), "restir_amd/host/strips_loopback_ranks is built by restir_amd/csrc/Makefile"
    r = subprocess.run([exe, "3", "200"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "strips_loopback_ranks ok (3 ranks)" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    # 8 modes on the library stream + 4 with the EAW filter, the tone map and the display gather on the denoise stream (rs_set_denoise_stream)
    assert r.stdout.count("== full frame over 6 frames: True") == 12, r.stdout[-3000:]
    assert r.stdout.count("on the denoise stream") == 4, r.stdout[-3000:]


def test_gbuffer_halo_setting_of_the_strip_driver(hip):
    """rs_strips_set_gbuffer_halo: 5 to 64 rows, never more than the shortest strip has; the value is the size of the messages, so a bad one
    is refused instead of sent (what it changes in a frame is compared with the full frame by strips_loopback_ranks)."""
    noop = lambda p, n, peer: None
    comm = hip.Comm(0, 2, noop, noop, None, None, stream_ordered=True)
    drv = hip.Strips(comm, 320, 100, [0, 40, 100])                    # strips of 40 and 60 rows
    try:
        for bad in (4, 65, 41):
            with pytest.raises(hip.RestirHipError):
                drv.set_gbuffer_halo(bad)
        drv.set_gbuffer_halo(32)
        drv.set_gbuffer_halo(40)
        drv.set_gbuffer_halo(5)
    finally:
        drv.destroy(); comm.destroy()


def test_strip_driver_eight_stream_ordered_ranks_on_one_gpu():
    """The same check with EIGHT ranks -- the split BASELINE's configs 4 and 5 name, and more processes than a one-GPU box lets a job put on
    its card, so here the ranks are threads of one process: 288 rows, 36 per strip, all 12 modes."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "restir_amd", "host", "strips_loopback_ranks")
    assert os.path.exists(exe), "restir_amd/host/strips_loopback_ranks is built by restir_amd/csrc/Makefile"
    r = subprocess.run([exe, "8", "500"], capture_output=True, text=True, timeout=560)
    assert r.returncode == 0 and "strips_loopback_ranks ok (8 ranks)" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("== full frame over 6 frames: True") == 12, r.stdout[-3000:]


def test_config4_4k_eight_strips_equal_full_frame(hip):
    """BASELINE config 4: the bench scene at 3840x2160 cut into 8 row strips with the 5-row reservoir halo -- the
    eight ranks are run one after the other on this GPU -- against the full-frame result, bit for bit."""
    import torch
    from restir_amd.tiling import HALO, HipBackend, strip_bounds
    sd = get_scene("sponza:1.0")
    W, H, N = 3840, 2160, 8
    scene = hip_scene(hip, sd)
    cam = hip.camera_update(sd.camera(W, H))
    full = HipBackend(hip, scene, cam, W, H)
    ranks = [HipBackend(hip, scene, cam, W, H) for _ in range(N)]
    bounds = [strip_bounds(H, N, r) for r in range(N)]
    for frame in range(2):
        full.gbuffer_render(0, H); full.phase_a(frame, 3, 0, H); full.phase_b(0, 3, 0, H); full.end_frame()
        for b, (y0, y1) in zip(ranks, bounds):
            b.gbuffer_render(max(0, y0 - HALO), min(H, y1 + HALO))
            b.phase_a(frame, 3, y0, y1)
        for r in range(N - 1):                                       # neighbours r (above) and r+1 (below)
            edge = bounds[r][1]
            down = ranks[r].halo_pack(edge - HALO, HALO); up = ranks[r + 1].halo_pack(edge, HALO)
            ranks[r + 1].halo_unpack(edge - HALO, HALO, down); ranks[r].halo_unpack(edge, HALO, up)
        for b, (y0, y1) in zip(ranks, bounds):
            b.phase_b(0, 3, y0, y1); b.end_frame()
        hip.synchronize()
        ref = full.image.cpu().numpy()
        got = np.concatenate([b.image[y0 * W:y1 * W].cpu().numpy() for b, (y0, y1) in zip(ranks, bounds)])
        assert np.isfinite(ref).all() and ref.max() > 0
        assert bits_equal(ref, got), (frame, radiance_stats(ref, got))
        assert full.restir.ray_count() == sum(b.restir.ray_count() for b in ranks)      # primary + shadow walks of the own rows


def test_full_size_properties(hip):
    """BASELINE config 3 size (1920x1080, spatiotemporal, Sponza-class 262 144 triangles): properties
    that do not need the oracle -- run-to-run determinism (no races), ray accounting, finite output,
    and agreement of a 64-row band with the oracle run on the same band's pixels is covered by the
    small-scene tests above."""
    sd = get_scene("sponza:1.0")
    W, H = 1920, 1080
    a = HipRenderer(hip, sd, W, H)
    b = HipRenderer(hip, sd, W, H, scene=a.scene)
    for frame in range(3):
        ia = a.frame(3); ib = b.frame(3)
        assert bits_equal(ia, ib)
        assert np.isfinite(ia).all()
        assert W * H <= a.rays <= 2 * W * H
    assert ia.mean() > 1e-3
    ga = a.gbuf.download()
    ids = ga["prim_id"][ga["frame_idx"] ^ 1]
    assert (ids >= -2).all() and (ids < len(sd.materials)).all()


def test_config3_full_size_orbit_of_64_frames(hip):
    """BASELINE config 3, second half: the 64-frame camera orbit (runCuda animateCamera, reprojection through devMotion every
    frame) at 1920x1080 on the full Sponza-class scene.  Frames enqueued without a host synchronisation (they overlap on the
    auxiliary streams) equal the synchronous run bit for bit -- images along the way, the final reservoirs and G-buffer -- and
    the temporal history is really in use (M grows beyond the 32 candidates of a single frame)."""
    import torch
    from restir_amd.scenes import orbit_position
    sd = get_scene("sponza:1.0")
    W, H, frames = 1920, 1080, 64
    scene = hip_scene(hip, sd)

    def run(overlapped):
        h = HipRenderer(hip, sd, W, H, scene=scene)
        keep = []
        hip.set_sync(not overlapped)
        try:
            for frame in range(frames):
                h.set_camera_position(orbit_position(sd.camera_args["position"], frame, radius=1.0))
                h.gbuf.render(h.scene, h.cam)
                h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, h.looper, 3)
                h.looper += 1
                if frame in (0, 17, 40, 63):
                    keep.append(h.image.clone())
                h.gbuf.update(h.cam)
            hip.synchronize(); torch.cuda.synchronize()
        finally:
            hip.set_sync(True)
        resv = h.restir.download(1)
        return [t.cpu().numpy() for t in keep], resv, h.gbuf.download()

    ia, ra, ga = run(True)
    ib, rb, gb = run(False)
    for a, b in zip(ia, ib):
        assert bits_equal(a, b)
        assert np.isfinite(a).all() and a.mean() > 1e-3
    assert ra.tobytes() == rb.tobytes()
    assert bits_equal(ga["depth"][0], gb["depth"][0]) and bits_equal(ga["depth"][1], gb["depth"][1])
    assert not bits_equal(ia[0], ia[-1])                               # the camera moved
    m = ra["numSamples"]
    assert m.max() == 32 * 20 and (m > 32).mean() > 0.3                # preClampedMerge<20>: history up to 19 x 32 on top of 32


def test_config5_bistro_class_with_eaw(hip):
    """BASELINE config 5 on one GPU: the Bistro-class scene at its stated size (2.83 M triangles, 10 240 emissive ones -- the RIS
    kernel that reads the light table from global memory), 1920x1080, spatiotemporal ReSTIR-DI with the 5-level EAW filter in the
    loop (runCuda's order: render, ReSTIRDirect, filter, update).  Full size: two independent runs, one of them with overlapped
    frames, agree bit for bit (radiance, filtered image, reservoirs); everything finite; ray accounting; the filter changes the
    image.  Oracle parity with the filter in the loop: the same pipeline on the scene at 5 % size, 240x136, radiance bit-exact
    and the filtered image within the EAW tolerance (expf)."""
    import torch
    sd = get_scene("bistro:1.0")
    assert 2.7e6 < sd.num_prims < 2.9e6
    lights = int((sd.materials["type"][sd.material_ids] == 4).sum())
    assert lights == 10240
    W, H = 1920, 1080
    scene = hip_scene(hip, sd)

    def run(overlapped):
        h = HipRenderer(hip, sd, W, H, scene=scene)
        f = hip.EAWFilter(W, H, 5)
        out = torch.zeros_like(h.image)
        res = torch.empty_like(h.image)
        hip.set_sync(not overlapped)
        rays = []
        try:
            for frame in range(3):
                h.gbuf.render(h.scene, h.cam)
                h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, h.looper, 3)
                h.looper += 1
                p = f.filter(out.data_ptr(), h.image.data_ptr(), h.gbuf, h.cam)
                hip.hip_memcpy_d2d(res.data_ptr(), p, res.numel() * 4)
                h.gbuf.update(h.cam)
                if not overlapped:
                    rays.append(h.restir.ray_count())
            hip.synchronize(); torch.cuda.synchronize()
        finally:
            hip.set_sync(True)
        f.destroy()
        return h.image.cpu().numpy(), res.cpu().numpy(), h.restir.download(1), rays

    ia, fa, ra, _ = run(True)
    ib, fb, rb, rays = run(False)
    assert bits_equal(ia, ib) and bits_equal(fa, fb) and ra.tobytes() == rb.tobytes()
    assert np.isfinite(ia).all() and np.isfinite(fa).all() and ia.mean() > 1e-3
    assert np.abs(fa - ia).max() > 1e-3                                 # the filter did something
    assert all(W * H <= r <= 2 * W * H for r in rays)                   # a shading ray per pixel + a shadow ray per shaded pixel
    del scene

    # the same loop against the oracle, on a scene and a frame the oracle finishes in seconds
    sd = get_scene("bistro:0.05")
    w, h_ = 240, 136
    o = OracleRenderer(sd, w, h_)
    g = HipRenderer(hip, sd, w, h_)
    f = hip.EAWFilter(w, h_, 5)
    out = torch.zeros_like(g.image)
    for frame in range(3):
        o.gbuf.render(o.scene, o.cam); g.gbuf.render(g.scene, g.cam)
        o.restir.direct(o.scene, o.cam, o.gbuf, o.image, 0, frame, 3)
        g.restir.direct(g.scene, g.cam, g.gbuf, g.image.data_ptr(), 0, frame, 3)
        ref = ob.eaw_filter(o.gbuf, o.cam, o.image)
        p = f.filter(out.data_ptr(), g.image.data_ptr(), g.gbuf, g.cam)
        hip.synchronize()
        res = torch.empty_like(g.image)
        hip.hip_memcpy_d2d(res.data_ptr(), p, res.numel() * 4)
        assert bits_equal(o.image, g.image.cpu().numpy()), frame
        assert np.allclose(ref, res.cpu().numpy(), rtol=1e-5, atol=1e-6), frame
        o.gbuf.update(o.cam); g.gbuf.update(g.cam)
    f.destroy()


def test_headless_viewer_drop_in(hip, tmp_path):
    """The reference's main()/runCuda() call sequence compiled against restir_compat.h
    (restir_amd/host/headless_viewer.cpp) produces the same RGBA8 frame as the oracle."""
    import os
    import subprocess
    from restir_amd import scenes
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "restir_amd", "host", "headless_viewer")
    assert os.path.exists(exe), "build it with make -C restir_amd/csrc"
    sd = get_scene("cornell")
    W, H, frames, reuse = 96, 64, 3, 1
    scenes.dump_scene(sd, str(tmp_path / "scene.bin"))
    subprocess.check_call([exe, str(tmp_path / "scene.bin"), str(W), str(H), str(frames), str(reuse), str(tmp_path / "out.ppm")])
    data = open(tmp_path / "out.ppm", "rb").read()
    header = f"P6\n{W} {H}\n255\n".encode()
    assert data.startswith(header)
    got = np.frombuffer(data[len(header):], np.uint8).reshape(H * W, 3)
    o = OracleRenderer(sd, W, H)
    for _ in range(frames):
        img = o.frame(reuse)
    ref = ob.send_image_to_pbo(img, W, H, 2, 1.0)[:, :3]
    diff = np.abs(ref.astype(np.int32) - got.astype(np.int32))
    assert diff.max() <= 1 and np.mean(diff > 0) <= 1e-3


def test_save_image_is_the_tone_mapped_frame_mirrored(hip, tmp_path):
    """saveImage(false) (src/main.cpp:105-144): tone map + gamma per pixel, stored at (width - 1 - x, y), clamped, x 255, truncated --
    the bytes of sendImageToPBO with scale 1, mirrored in x, as an 8-bit RGB PNG; compared with the oracle's bytes (<= 1 LSB on <= 1e-3
    of them, the display conversion's tolerance), through the C ABI and through the viewer's ".png" output."""
    import os, subprocess, torch
    from tests.common import read_png_rgb
    from restir_amd import scenes
    sd = get_scene("cornell")
    W, H, frames = 96, 64, 3
    h = HipRenderer(hip, sd, W, H)
    o = OracleRenderer(sd, W, H)
    for _ in range(frames):
        img = o.frame(1); h.frame(1)
    for mode in (0, 1, 2):
        hip.save_image(tmp_path / ("m%d.png" % mode), h.image.data_ptr(), W, H, mode)
        got = read_png_rgb(tmp_path / ("m%d.png" % mode))
        ref = ob.send_image_to_pbo(img, W, H, mode, 1.0)[:, :3].reshape(H, W, 3)[:, ::-1]
        diff = np.abs(ref.astype(np.int32) - got.astype(np.int32))
        assert diff.max() <= 1 and np.mean(diff > 0) <= 1e-3, mode
        # saveImage(true): the same 8-bit picture through Image::saveJPG's writer (byte-exact against the reference's own file in
        # tests/test_host_and_abi.py): the file of rs_save_image_jpg is rs_write_jpg of the PNG's pixels
        hip.save_image_jpg(tmp_path / ("m%d.jpg" % mode), h.image.data_ptr(), W, H, mode)
        hip.write_jpg(tmp_path / ("m%d_from_png.jpg" % mode), got)
        a, b = open(tmp_path / ("m%d.jpg" % mode), "rb").read(), open(tmp_path / ("m%d_from_png.jpg" % mode), "rb").read()
        assert a == b and a[:4] == b"\xff\xd8\xff\xe0" and a[-2:] == b"\xff\xd9"
        pbo = torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda")          # and exactly the library's own PBO bytes, mirrored
        hip.copy_image_to_pbo(pbo.data_ptr(), h.image.data_ptr(), W, H, mode, 1.0)
        assert np.array_equal(pbo.cpu().numpy()[:, :3].reshape(H, W, 3)[:, ::-1], got)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "restir_amd", "host", "headless_viewer")
    scenes.dump_scene(sd, str(tmp_path / "scene.bin"))
    subprocess.check_call([exe, str(tmp_path / "scene.bin"), str(W), str(H), str(frames), "1", str(tmp_path / "shot.png")])
    assert np.array_equal(read_png_rgb(tmp_path / "shot.png"), read_png_rgb(tmp_path / "m2.png"))      # Settings::toneMapping = ACES


def test_pbo_registration_without_a_gl_context_fails_cleanly(hip):
    """cudaGLRegisterBufferObject's replacement (rs_pbo_register -> hipGraphicsGLRegisterBuffer) on a node without OpenGL: an error
    code and a message, no crash, and the library keeps working."""
    p = C.c_void_p()
    rc = hip.lib().rs_pbo_register(1, C.byref(p))
    assert rc != 0 and not p.value
    assert b"rs_pbo_register" in hip.lib().rs_last_error()
    assert hip.lib().rs_pbo_unmap(None) != 0 and hip.lib().rs_pbo_unregister(None) == 0
    sd = get_scene("cornell")
    h = HipRenderer(hip, sd, 32, 24)
    assert np.isfinite(h.frame(0)).all()
