#!/usr/bin/env python3
"""tools/bench_denoisers.py -- 1080p timings of the two denoisers of src/denoiser.cu on the bench scene's G-buffer:
LeveledEAWFilter (5 a-trous levels) and SpatioTemporalFilter (SVGF).  Algorithmic bytes per pixel:
EAW level = read colour 12 + G-buffer {id 4, normal 12, depth 4} + write 12 = 44 B; SVGF level adds variance in/out/filtered
(12 B) and the 3x3 variance pre-filter (8 B); temporal accumulation reads colour 12 + motion 4 + 2 x {id 4, normal 12} +
history colour/moment 24 and writes 24 = 96 B; variance estimate 12 + 4 B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from restir_amd import capi, scenes
from restir_amd.tiling import HipBackend

W, H = 1920, 1080
capi.init(0)
sd = scenes.bistro_class(seed=2, scale=1.0) if "--bistro" in sys.argv else scenes.sponza_class(seed=1, scale=1.0)      # --bistro: config 5's scene
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
b = HipBackend(capi, scene, cam, W, H)
capi.set_sync(False)
for i in range(3):
    b.gbuffer_render(0, H); b.phase_a(i, 3, 0, H); b.phase_b(0, 3, 0, H); b.restir.end_frame()
torch.cuda.synchronize()
N = W * H


def span(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def timed(fn, reps=100, load_s=0.25):
    """(us per call after load_s seconds of continuous calls, us per call over the first 23 calls).  A denoiser runs every frame of a
    real-time renderer, so the figure that counts is the one under continuous load; the first calls after the host-side set-up run on
    a GPU whose clocks are still coming up from idle (20 % slower for these issue-bound kernels) and are printed next to it."""
    import time
    for _ in range(3):
        fn()
    first = span(fn, 20)
    t0 = time.time()
    while time.time() - t0 < load_s:
        span(fn, 20)
    return span(fn, reps), first


eaw = capi.EAWFilter(W, H, 5)
out = torch.zeros_like(b.image)
state = {"p": out.data_ptr()}
t, first = timed(lambda: state.update(p=eaw.filter(state["p"], b.image.data_ptr(), b.gbuf, cam)))
bytes_eaw = N * (5 * 44 + 16 + 12)            # + the position plane: read depth/id 8 + ... write 12
print("LeveledEAWFilter   %.1f us / frame, algorithmic %.0f MB -> %.0f GB/s (%.2f of 8 TB/s); first 23 calls of the process %.1f us (%.2f)"
      % (t, bytes_eaw / 1e6, bytes_eaw / t / 1e3, bytes_eaw / t / 1e3 / 8000, first, bytes_eaw / first / 1e3 / 8000))
if len(sys.argv) > 1 and sys.argv[1] == "eaw":        # tools/profile_eaw.sh: the EAW filter only
    sys.exit(0)
svgf = capi.SVGFFilter(W, H, 5)
def svgf_frame():
    svgf.filter(b.image.data_ptr(), b.gbuf, cam); svgf.next_frame()
t, first = timed(svgf_frame)
bytes_svgf = N * (96 + 16 + 5 * (44 + 12 + 8) + 20)
print("SpatioTemporalFilter %.1f us / frame, algorithmic %.0f MB -> %.0f GB/s (%.2f of 8 TB/s); first 23 calls %.1f us (%.2f)"
      % (t, bytes_svgf / 1e6, bytes_svgf / t / 1e3, bytes_svgf / t / 1e3 / 8000, first, bytes_svgf / first / 1e3 / 8000))
