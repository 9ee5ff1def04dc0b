"""CPU, world_size 2 and 3 over gloo: the row-strip decomposition of restir_amd/tiling.py (strip
bounds, 5-row reservoir halo exchange, history all-gather for a moving camera) reproduces the
single-process full-frame result bit for bit.  The per-rank compute is the CPU oracle behind the
same backend interface the HIP backend implements (tests/common.py OracleBackend)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

W, H, FRAMES = 96, 64, 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


BALANCED = {3: [(0, 8), (8, 45), (45, 64)]}     # uneven (cost-balanced style) strips: still exact


def _worker(rank, world, port, moving, outdir, balanced=False, calibrated=False, denoise=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as ob
    from restir_amd.scenes import orbit_position
    from restir_amd.tiling import StripRenderer
    from tests.common import OracleBackend, get_scene, oracle_scene
    sd = get_scene("sponza:0.02")
    cam = ob.camera_update(sd.camera(W, H))
    backend = OracleBackend(oracle_scene(sd), cam, W, H)
    bounds = BALANCED[world] if balanced else None
    if calibrated:                                  # what bench.py does for N > 1: measure, all-gather, rebalance; then fresh buffers
        from restir_amd.tiling import calibrate_bounds
        bounds = calibrate_bounds(backend, world, rank, H, dist, lambda: None, rounds=2, frames=1, denoise=denoise, min_rows=32 if denoise else 8)
        backend = OracleBackend(backend.scene, cam, W, H)
        np.save(os.path.join(outdir, f"bounds_{rank}.npy"), np.array(bounds))
    r = StripRenderer(backend, world, rank, H, dist=dist, share_history=moving, bounds=bounds)
    for frame in range(FRAMES):
        if moving:
            p = orbit_position(sd.camera_args["position"], frame, radius=0.6)
            for i in range(3):
                cam.position[i] = float(p[i])
            ob.camera_update(cam)
        r.frame(3, denoise=denoise)
    np.save(os.path.join(outdir, f"strip_{rank}.npy"), backend.image[r.y0 * W:r.y1 * W])
    if denoise:
        np.save(os.path.join(outdir, f"filtered_{rank}.npy"), r.filtered[r.y0 * W:r.y1 * W])
    dist.barrier()
    dist.destroy_process_group()


def _reference(moving):
    from oracle import binding as ob
    from restir_amd.scenes import orbit_position
    from tests.common import OracleRenderer, get_scene
    sd = get_scene("sponza:0.02")
    o = OracleRenderer(sd, W, H)
    for frame in range(FRAMES):
        if moving:
            o.set_camera_position(orbit_position(sd.camera_args["position"], frame, radius=0.6))
        img = o.frame(3)
    return img.copy()


@pytest.mark.parametrize("world,moving", [(2, False), (2, True), (3, True)])
def test_strips_equal_full_frame(tmp_path, world, moving):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, moving, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"strip_{r}.npy") for r in range(world)])
    ref = _reference(moving)
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_uneven_strips_equal_full_frame(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(3, port, True, str(tmp_path), True), nprocs=3, join=True)
    got = np.concatenate([np.load(tmp_path / f"strip_{r}.npy") for r in range(3)])
    assert np.array_equal(got.view(np.uint32), _reference(True).view(np.uint32))


def test_calibrated_strips_equal_full_frame(tmp_path):
    """Strip heights chosen by measurement (timing, all-gather, rebalance -- identical on every rank); the frames
    rendered afterwards with fresh buffers equal the full-frame render."""
    port = _free_port()
    mp.spawn(_worker, args=(3, port, True, str(tmp_path), False, True), nprocs=3, join=True)
    b = [np.load(tmp_path / f"bounds_{r}.npy") for r in range(3)]
    assert all(np.array_equal(b[0], x) for x in b) and b[0][0][0] == 0 and b[0][-1][1] == H
    got = np.concatenate([np.load(tmp_path / f"strip_{r}.npy") for r in range(3)])
    assert np.array_equal(got.view(np.uint32), _reference(True).view(np.uint32))


def test_eight_calibrated_strips_equal_full_frame(tmp_path):
    """World size 8, the split BASELINE's multi-GPU configs name: calibrate_bounds' eight-way all-gather and rebalancing (identical bounds on
    every rank, no strip below 8 rows), two neighbours for six of the ranks, the history exchange of a moving camera among eight."""
    port = _free_port()
    mp.spawn(_worker, args=(8, port, True, str(tmp_path), False, True), nprocs=8, join=True)
    b = [np.load(tmp_path / f"bounds_{r}.npy") for r in range(8)]
    assert all(np.array_equal(b[0], x) for x in b) and b[0][0][0] == 0 and b[0][-1][1] == H and len(b[0]) == 8
    assert min(y1 - y0 for y0, y1 in b[0]) >= 8
    got = np.concatenate([np.load(tmp_path / f"strip_{r}.npy") for r in range(8)])
    assert np.array_equal(got.view(np.uint32), _reference(True).view(np.uint32))


def test_eaw_on_strips_equals_full_frame_filter(tmp_path):
    """BASELINE config 5's denoiser on a tiled framebuffer: five a-trous levels per strip with the 2 << level border rows of each
    level's input (and 32 G-buffer rows once) exchanged between neighbours reproduce the full-frame LeveledEAWFilter."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, True, str(tmp_path), False, False, True), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / f"filtered_{r}.npy") for r in range(2)])
    from oracle import binding as ob
    from restir_amd.scenes import orbit_position
    from tests.common import OracleRenderer, get_scene
    sd = get_scene("sponza:0.02")
    o = OracleRenderer(sd, W, H)
    for frame in range(FRAMES):
        o.set_camera_position(orbit_position(sd.camera_args["position"], frame, radius=0.6))
        o.gbuf.render(o.scene, o.cam)
        o.restir.direct(o.scene, o.cam, o.gbuf, o.image, 0, o.looper, 3)
        o.looper += 1
        ref = ob.eaw_filter(o.gbuf, o.cam, o.image).copy()          # denoise before GBuffer::update, as runCuda does
        o.gbuf.update(o.cam)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert np.abs(ref - o.image).max() > 1e-3


def test_calibrated_strips_with_the_filter(tmp_path):
    """What `bench.py --config 5 --gpus N` does: strip heights calibrated on frames that include the filter, never below the 32 rows its
    levels reach into the neighbours; the filtered strips equal the full-frame LeveledEAWFilter."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, False, str(tmp_path), False, True, True), nprocs=2, join=True)
    b = [np.load(tmp_path / f"bounds_{r}.npy") for r in range(2)]
    assert np.array_equal(b[0], b[1]) and min(y1 - y0 for y0, y1 in b[0]) >= 32
    got = np.concatenate([np.load(tmp_path / f"filtered_{r}.npy") for r in range(2)])
    from oracle import binding as ob
    from tests.common import OracleRenderer, get_scene
    o = OracleRenderer(get_scene("sponza:0.02"), W, H)
    for frame in range(FRAMES):
        o.gbuf.render(o.scene, o.cam)
        o.restir.direct(o.scene, o.cam, o.gbuf, o.image, 0, o.looper, 3)
        o.looper += 1
        ref = ob.eaw_filter(o.gbuf, o.cam, o.image).copy()
        o.gbuf.update(o.cam)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_rebalance_bounds():
    from restir_amd.tiling import HALO, StripRenderer, rebalance_bounds, strip_bounds
    for height, world in ((1080, 8), (1080, 2), (2160, 8), (64, 3)):
        b = [strip_bounds(height, world, r) for r in range(world)]
        rng = np.random.default_rng(world)
        for _ in range(5):
            t = rng.uniform(0.2, 2.0, world)
            nb = rebalance_bounds(b, t, height)
            assert nb[0][0] == 0 and nb[-1][1] == height
            assert all(nb[i][1] == nb[i + 1][0] for i in range(world - 1))
            assert min(y1 - y0 for y0, y1 in nb) >= max(HALO, 8)
            b = nb
    for t in ([5.0, 1.0, 1.0, 1.0], [1.0, 1.0, 1.0, 9.0]):           # the filter's floor: no strip below 32 rows however skewed the costs
        nb = rebalance_bounds([strip_bounds(1080, 4, r) for r in range(4)], t, 1080, min_rows=32)
        assert min(y1 - y0 for y0, y1 in nb) >= 32 and nb[0][0] == 0 and nb[-1][1] == 1080
    nb = rebalance_bounds([(0, 32), (32, 64)], [9.0, 1.0], 64, min_rows=32)
    assert nb == [(0, 32), (32, 64)]
    # a strip that costs more per row gets fewer rows; equal costs keep an even split (up to the 8-row quantum)
    nb = rebalance_bounds([(0, 540), (540, 1080)], [2.0, 1.0], 1080)
    assert nb[0][1] < 540
    nb = rebalance_bounds([(0, 540), (540, 1080)], [1.0, 1.0], 1080)
    assert abs(nb[0][1] - 540) <= 8
    with pytest.raises(ValueError):
        StripRenderer(None, 2, 0, 64, bounds=[(0, 30), (32, 64)])


def test_strip_bounds_cover_and_balance():
    from restir_amd.tiling import strip_bounds
    for height in (1080, 2160, 64, 135):
        for world in (1, 2, 3, 4, 8):
            b = [strip_bounds(height, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == height
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [y1 - y0 for y0, y1 in b]
            assert max(sizes) - min(sizes) <= 1
