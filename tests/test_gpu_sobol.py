"""GPU parity tests of the Sobol sampler branch (src/sampler.h:9-36, SAMPLER_USE_SOBOL true; table upload src/scene.cpp:500-506;
`State::looper = (State::looper + 1) % SobolSampleNum`, src/restir.cu:441-445): librestir_hip with rs_scene_set_sample_sequence
against the oracle with the same table, bit for bit -- RIS, temporal and spatial reuse, PT-direct, the multi-bounce kernels, both
RIS table locations, a textured / environment-lit scene, the looper wrap, strips, overlapped launches.

The table is the build's own (restir_amd/sobol.py; the reference's file is absent from its repository), so this is parity against
the oracle only -- stated in DESIGN.md."""
import numpy as np
import pytest

from oracle import binding as ob
from restir_amd import sobol
from tests.common import HipRenderer, OracleRenderer, bits_equal, get_scene, hip_scene, radiance_stats
from tests.test_gpu_parity import SCENES, _compare_reservoirs, _gi_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def table():
    return sobol.sobol_table()


@pytest.fixture(autouse=True)
def exact_libm():
    ob.set_libm_mode(1)            # cos / sin / atan2 correctly rounded on both sides: every bit must agree
    yield
    ob.set_libm_mode(0)


@pytest.mark.parametrize("name", list(SCENES))
@pytest.mark.parametrize("reuse", [0, 1, 2, 3])
def test_sobol_restir_direct_bit_exact(hip, table, name, reuse):
    """ReSTIRDirect in every reuse mode across the wrap of the looper (frames 9998, 9999, 0, 1): radiance, ray counts, the stored and
    the published reservoirs."""
    sd = get_scene(name)
    W, H = SCENES[name]
    o = OracleRenderer(sd, W, H, sobol=table)
    h = HipRenderer(hip, sd, W, H, sobol=table)
    o.looper = h.looper = 9998
    for frame in range(4):
        a = o.frame(reuse); b = h.frame(reuse)
        assert o.rays == h.rays, (frame, o.rays, h.rays)
        assert bits_equal(a, b), (frame, radiance_stats(a, b))
        _compare_reservoirs(o.restir.last, h.restir.download(1))
        if reuse & 2:
            _compare_reservoirs(o.restir.temp, h.restir.download(2))
    assert o.looper == h.looper == 2
    # a different sequence from the default engine's
    d = HipRenderer(hip, sd, W, H, scene=h.scene, sobol=None)
    d.scene.set_sample_sequence(None)
    d.looper = 9998
    for frame in range(4):
        c = d.frame(reuse)
    assert not bits_equal(b, c)
    ref = OracleRenderer(sd, W, H)
    ref.looper = 9998
    for frame in range(4):
        e = ref.frame(reuse)
    assert bits_equal(c, e)                                            # and the switch back is the default engine again


@pytest.mark.parametrize("table_in", ["lds", "global"])
def test_sobol_both_ris_table_locations(hip, table, table_in):
    sd = get_scene("sponza:0.03")
    W, H = 160, 96
    hip.set_ris_table_pixels(0 if table_in == "lds" else 1 << 30)
    try:
        o = OracleRenderer(sd, W, H, sobol=table)
        h = HipRenderer(hip, sd, W, H, sobol=table)
        for frame in range(3):
            a = o.frame(3); b = h.frame(3)
            assert bits_equal(a, b), (frame, radiance_stats(a, b))
    finally:
        hip.set_ris_table_pixels(384 * 1024)


def test_sobol_textured_and_environment_lit_scene(hip, table):
    """The ENV / TEX instantiations: the environment map as the sampler's last light (scene.h:400-403) draws through the same
    sampler (r.z, r.w of the candidate's sample4D)."""
    sd = get_scene("cornell_textured")
    W, H = 160, 120
    o = OracleRenderer(sd, W, H, sobol=table)
    h = HipRenderer(hip, sd, W, H, sobol=table)
    for frame in range(3):
        a = o.frame(3); b = h.frame(3)
        assert o.rays == h.rays and bits_equal(a, b), (frame, radiance_stats(a, b))
        _compare_reservoirs(o.restir.last, h.restir.download(1))
    o = OracleRenderer(sd, W, H, sobol=table); h = HipRenderer(hip, sd, W, H, sobol=table)
    a = o.frame(0, use_reservoir=False); b = h.frame(0, use_reservoir=False)
    assert o.rays == h.rays and bits_equal(a, b)


@pytest.mark.parametrize("name", list(SCENES))
def test_sobol_path_trace_direct(hip, table, name):
    sd = get_scene(name)
    W, H = SCENES[name]
    o = OracleRenderer(sd, W, H, sobol=table)
    h = HipRenderer(hip, sd, W, H, sobol=table)
    for it in range(3):
        a = o.frame(0, use_reservoir=False, iteration=it); b = h.frame(0, use_reservoir=False, iteration=it)
        assert o.rays == h.rays
        assert bits_equal(a, b), (it, radiance_stats(a, b))


@pytest.mark.parametrize("name", ["cornell", "cornell_glass", "sponza:0.03"])
def test_sobol_multi_bounce_kernels(hip, table, name):
    """pathTrace, pathTraceIndirect and ReSTIRIndirect draw a different number of values per pixel (paths end at different depths):
    the per-lane table position must follow each path."""
    import torch
    sd = _gi_scene(name)
    W, H = 96, 64
    o = OracleRenderer(sd, W, H, sobol=table)
    h = HipRenderer(hip, sd, W, H, sobol=table)
    od = np.zeros((W * H, 3), np.float32); oi = np.zeros((W * H, 3), np.float32)
    hd = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda"); hi = torch.zeros_like(hd)
    for frame, depth in enumerate((1, 3, 5)):
        looper = 9999 if frame == 2 else frame                         # the table's last row: long paths read on into the guard
        ra = ob.path_trace(o.scene, o.cam, od, oi, frame, looper, depth)
        rb = hip.path_trace(h.scene, h.cam, hd.data_ptr(), hi.data_ptr(), frame, looper, depth)
        assert ra == rb, (frame, ra, rb)
        assert bits_equal(od, hd.cpu().numpy()) and bits_equal(oi, hi.cpu().numpy()), (frame, radiance_stats(oi, hi.cpu().numpy()))
    oi[:] = 0; hi.zero_()
    for frame, depth in enumerate((2, 4)):
        ra = ob.pt_indirect(o.scene, o.cam, oi, frame, 7 + frame, depth)
        rb = hip.path_trace_indirect(h.scene, h.cam, hi.data_ptr(), frame, 7 + frame, depth)
        assert ra == rb and bits_equal(oi, hi.cpu().numpy()), (frame, radiance_stats(oi, hi.cpu().numpy()))
    oi[:] = 0; hi.zero_()
    for frame in range(3):
        o.gbuf.render(o.scene, o.cam); h.gbuf.render(h.scene, h.cam)
        ra = o.restir.indirect(o.scene, o.cam, o.gbuf, oi, 0, frame, 1, 4)
        rb = h.restir.indirect(h.scene, h.cam, h.gbuf, hi.data_ptr(), 0, frame, 1, 4)
        assert ra == rb and bits_equal(oi, hi.cpu().numpy()), (frame, radiance_stats(oi, hi.cpu().numpy()))
        o.gbuf.update(o.cam); h.gbuf.update(h.cam)


def test_sobol_looper_outside_the_table_is_refused(hip, table):
    sd = get_scene("cornell")
    W, H = 64, 48
    h = HipRenderer(hip, sd, W, H, sobol=table[:100])                  # a 100-row table
    h.looper = 99
    h.frame(3)
    assert h.looper == 0
    h.looper = 100
    with pytest.raises(RuntimeError, match="looper outside the Sobol table"):
        h.frame(3)
    h.looper = -1
    with pytest.raises(RuntimeError, match="looper outside the Sobol table"):
        h.frame(0, use_reservoir=False)
    with pytest.raises(RuntimeError, match="numSamples x 200"):
        h.scene.set_sample_sequence(np.zeros((10, 64), np.uint32))
    h.scene.set_sample_sequence(None)
    h.looper = 12345
    h.frame(3)                                                         # the default engine takes any looper


def test_sqrt_of_uniform_on_every_value_of_the_sobol_sampler(hip):
    """sqrt_of_uniform (rs_surface.h) against the exactly rounded sqrtf on 0 and every float in [2^-32, 1]: all that
    (float)r * 2^-32 can be."""
    import ctypes as C
    bad = C.c_ulonglong(1)
    hip.check(hip.lib().rs_debug_sqrt_of_unit_floats_mismatches(C.byref(bad)))
    assert bad.value == 0, bad.value


def test_sobol_strips_equal_full_frame(hip, table):
    """Row strips in Sobol mode: phase B of a strip resumes each pixel's sampler from the scramble phase A left and the table row
    of the frame's looper."""
    import torch
    sd = get_scene("sponza:0.03")
    W, H = 160, 96
    full = HipRenderer(hip, sd, W, H, sobol=table)
    ranks = [HipRenderer(hip, sd, W, H, scene=full.scene, sobol=table) for _ in range(2)]
    bounds = [(0, 40), (40, H)]
    halo = 5
    for frame in range(3):
        ref = full.frame(3)
        for r, (y0, y1) in zip(ranks, bounds):
            r.gbuf.render(r.scene, r.cam, y0, y1)
            r.restir.phase_a(r.scene, r.cam, r.gbuf, r.looper, 3, y0, y1)
        nr, ng = ranks[0].restir.halo_bytes(halo), ranks[0].gbuf.rows_bytes(halo)
        up = torch.empty(nr + ng, dtype=torch.uint8, device="cuda"); down = torch.empty(nr + ng, dtype=torch.uint8, device="cuda")
        ranks[0].restir.halo_pack(bounds[0][1] - halo, halo, down.data_ptr())
        ranks[0].gbuf.rows_pack(0, bounds[0][1] - halo, halo, down.data_ptr() + nr)
        ranks[1].restir.halo_pack(bounds[1][0], halo, up.data_ptr())
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


def test_sobol_overlapped_frames_equal_synchronous_frames(hip, table):
    """The overlapped mode in Sobol mode (frames in flight keep their own looper: phase B's table row is the frame's)."""
    import torch
    from restir_amd.scenes import orbit_position
    sd = get_scene("sponza:0.2")
    W, H, frames = 640, 360, 10
    scene = hip_scene(hip, sd)

    def run(overlapped):
        h = HipRenderer(hip, sd, W, H, scene=scene, sobol=table)
        keep = []
        hip.set_sync(not overlapped)
        try:
            for frame in range(frames):
                h.set_camera_position(orbit_position(sd.camera_args["position"], frame, radius=0.5))
                h.gbuf.render(h.scene, h.cam)
                h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, 9995 + frame if frame < 5 else frame - 5, 3)
                keep.append(h.image.clone())
                h.gbuf.update(h.cam)
            hip.synchronize(); torch.cuda.synchronize()
        finally:
            hip.set_sync(True)
        return [t.cpu().numpy() for t in keep], h.restir.download(1)

    ia, ra = run(True)
    ib, rb = run(False)
    for a, b in zip(ia, ib):
        assert bits_equal(a, b)
    assert ra.tobytes() == rb.tobytes()
    o = OracleRenderer(sd, W, H, sobol=table)
    for frame in range(2):                                             # and the first frames against the oracle
        o.set_camera_position(orbit_position(sd.camera_args["position"], frame, radius=0.5))
        o.looper = 9995 + frame
        assert bits_equal(o.frame(3), ib[frame]), frame
