"""GPU parity tests of the tile splitting of the closest-hit kernels (restir_amd/csrc/rs_tilesplit.h): a tile whose packet walk
was long the last time the same launch ran is traced by four waves of 16 rays instead of one of 64.  Which rays share a wave
changes no result, so with ANY threshold -- 8 union nodes: practically every tile of every frame after the first is split -- the
frames, reservoirs, G-buffer planes and ray counts must equal the oracle's and the unsplit run's bit for bit, synchronous and
with the frames overlapped, separate and fused launches, full frames and row ranges."""
import numpy as np
import pytest

from oracle import binding as ob
from tests.common import HipRenderer, OracleRenderer, bits_equal, get_scene, hip_scene, radiance_stats

pytestmark = pytest.mark.gpu


@pytest.fixture
def exact_libm():
    ob.set_libm_mode(1)
    yield
    ob.set_libm_mode(0)


def _same_gbuffer(a, b):
    assert np.array_equal(a["motion"], b["motion"]) and bits_equal(a["albedo"], b["albedo"])
    for i in range(2):
        assert np.array_equal(a["prim_id"][i], b["prim_id"][i])
        assert bits_equal(a["normal"][i], b["normal"][i]) and bits_equal(a["depth"][i], b["depth"][i])


def _set(hip, threshold):
    from restir_amd import capi
    capi.check(capi.lib().rs_set_tile_split(int(threshold)))


@pytest.mark.parametrize("name,size", [("cornell", (128, 96)), ("sponza:0.1", (240, 136))])
@pytest.mark.parametrize("threshold", [8, 96])
def test_split_tiles_equal_oracle(hip, exact_libm, name, size, threshold):
    from restir_amd.scenes import orbit_position
    sd = get_scene(name)
    W, H = size
    _set(hip, threshold)
    try:
        o = OracleRenderer(sd, W, H)
        h = HipRenderer(hip, sd, W, H)
        for frame in range(6):
            if frame >= 3:
                p = orbit_position(sd.camera_args["position"], frame, radius=0.4)
                o.set_camera_position(p); h.set_camera_position(p)
            a = o.frame(3)
            b = h.frame(3)
            st = radiance_stats(a, b)
            assert st["bit_mismatch"] == 0, (frame, st)
            assert o.rays == h.rays, (frame, o.rays, h.rays)
        g = h.gbuf.download()
        f = g["frame_idx"] ^ 1                                    # the planes rendered last (update() flipped the index)
        assert np.array_equal(o.gbuf.prim_id[f], g["prim_id"][f]) and np.array_equal(o.gbuf.motion, g["motion"])
        assert bits_equal(o.gbuf.albedo, g["albedo"]) and bits_equal(o.gbuf.normal[f], g["normal"][f]) and bits_equal(o.gbuf.depth[f], g["depth"][f])
    finally:
        _set(hip, 768)


@pytest.mark.parametrize("fused", [False, True])
def test_split_tiles_in_overlapped_frames(hip, fused):
    """Twelve frames of an orbiting camera enqueued without a host synchronisation (three chains in flight, every chain stream with
    its own hints), with a second partial render in some frames: threshold 8 = threshold 0 (off), images, reservoirs, G-buffer, rays."""
    import torch
    from restir_amd.scenes import orbit_position
    sd = get_scene("sponza:0.2")
    W, H, frames = 640, 360, 12
    scene = hip_scene(hip, sd)

    def run(threshold):
        _set(hip, threshold)
        h = HipRenderer(hip, sd, W, H, scene=scene)
        images, rays = [], []
        hip.set_sync(False)
        hip.set_side_stream(3 if fused else 1)
        try:
            for frame in range(frames):
                h.set_camera_position(orbit_position(sd.camera_args["position"], frame // 2, radius=0.5))     # every view twice: fresh hints, then stale ones
                h.gbuf.render(h.scene, h.cam)
                if frame % 5 == 3:
                    h.gbuf.render(h.scene, h.cam, 0, H // 2)
                h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, h.looper, 3)
                h.looper += 1
                images.append(h.image.clone())
                h.gbuf.update(h.cam)
            hip.synchronize()
            torch.cuda.synchronize()
            rays = h.restir.ray_count()
        finally:
            hip.set_sync(True)
            hip.set_side_stream(4)
            _set(hip, 768)
        return dict(images=[t.cpu().numpy() for t in images], resv=h.restir.download(1), gbuf=h.gbuf.download(), rays=rays)

    a, b = run(-8), run(0)                 # negative: also for launches that overlap others
    for f in range(frames):
        assert bits_equal(a["images"][f], b["images"][f]), f
    assert a["rays"] == b["rays"]
    for k in a["resv"].dtype.names:
        assert bits_equal(a["resv"][k], b["resv"][k]), k
    _same_gbuffer(a["gbuf"], b["gbuf"])


def test_split_tiles_on_row_ranges(hip):
    """Strips: GBuffer::render and ReSTIRDirect's phases on a row range that starts and ends inside tiles, several frames."""
    from restir_amd.tiling import HipBackend
    sd = get_scene("sponza:0.1")
    W, H = 320, 203
    scene = hip_scene(hip, sd)
    cam = hip.camera_update(sd.camera(W, H))
    y0, y1 = 37, 171

    def run(threshold):
        _set(hip, threshold)
        try:
            b = HipBackend(hip, scene, cam, W, H)
            out = []
            for f in range(5):
                b.gbuffer_render(y0 - 5, y1 + 5)
                b.phase_a(f, 1, y0, y1)
                b.phase_b(0, 1, y0, y1)
                b.end_frame()
                out.append(b.image.cpu().numpy().reshape(H, W, 3)[y0:y1].copy())
            return out
        finally:
            _set(hip, 768)

    a, b = run(8), run(0)
    for x, y in zip(a, b):
        assert bits_equal(x, y)
    assert np.abs(a[-1]).sum() > 0


@pytest.mark.parametrize("threshold", [6, 3000])
def test_adaptive_split_of_small_overlapped_launches(hip, threshold):
    """A POSITIVE threshold splits overlapped launches of less than three rounds of wave slots at 4/3 of it, and only while heavy tiles
    are reported (rs_tile_split_prepare, mode 2): a site without any sleeps for 29 launches, probes twice, sleeps again.  100 overlapped
    frames of an orbiting camera (threshold 6: practically every tile is heavy, the sites stay awake; 3000: none is, they sleep and
    probe) equal the frames with the feature off -- images, reservoirs, G-buffer planes, ray counts."""
    import torch
    from restir_amd.scenes import orbit_position
    sd = get_scene("sponza:0.2")
    W, H, frames = 640, 360, 100
    scene = hip_scene(hip, sd)

    def run(thr):
        _set(hip, thr)
        h = HipRenderer(hip, sd, W, H, scene=scene)
        keep = {}
        hip.set_sync(False)
        try:
            for frame in range(frames):
                h.set_camera_position(orbit_position(sd.camera_args["position"], frame // 3, radius=0.5))
                h.gbuf.render(h.scene, h.cam)
                h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, h.looper, 3)
                h.looper += 1
                if frame % 9 == 0 or frame == frames - 1:
                    keep[frame] = h.image.clone()
                h.gbuf.update(h.cam)
                if frame % 25 == 24:
                    hip.synchronize()              # lets the reports arrive: later launches see fresh verdicts
            hip.synchronize()
            torch.cuda.synchronize()
            rays = h.restir.ray_count()
        finally:
            hip.set_sync(True)
            _set(hip, 768)
        return dict(images={k: t.cpu().numpy() for k, t in keep.items()}, resv=h.restir.download(1), gbuf=h.gbuf.download(), rays=rays)

    a, b = run(threshold), run(0)
    for f in a["images"]:
        assert bits_equal(a["images"][f], b["images"][f]), f
    assert a["rays"] == b["rays"]
    for k in a["resv"].dtype.names:
        assert bits_equal(a["resv"][k], b["resv"][k]), k
    _same_gbuffer(a["gbuf"], b["gbuf"])
