#!/usr/bin/env python3
"""tools/eaw_fused_cost.py [--bistro] -- what the reference's operation order costs the EAW filter: every a-trous level of LeveledEAWFilter at 1080p
with the taps in fused arithmetic (the default, rs_eaw_set_fused(f, 1)) and with every operation rounded separately in the reference's order
(rs_eaw_set_fused(f, 0), src/denoiser.cu:64-134), interleaved, HIP events around 50 launches per level after a warm-up under load."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from restir_amd import capi, scenes
from restir_amd.tiling import HipBackend

W, H = 1920, 1080
capi.init(0)
sd = scenes.bistro_class(seed=2, scale=1.0) if "--bistro" in sys.argv else scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
b = HipBackend(capi, scene, cam, W, H)
capi.set_sync(False)
for i in range(3):
    b.gbuffer_render(0, H); b.phase_a(i, 3, 0, H); b.phase_b(0, 3, 0, H); b.restir.end_frame()
b.gbuffer_render(0, H); b.phase_a(3, 3, 0, H); b.phase_b(0, 3, 0, H)
f = capi.EAWFilter(W, H, 5)
f.positions_rows(b.gbuf, cam, 0, H)
bufs = [torch.zeros_like(b.image), torch.zeros_like(b.image)]
torch.cuda.synchronize()


def level_us(level, reps=50):
    src = b.image if level == 0 else bufs[(level - 1) % 2]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        f.level_rows(bufs[level % 2].data_ptr(), src.data_ptr(), b.gbuf, level, 0, H)
    e0.record()
    for _ in range(reps):
        f.level_rows(bufs[level % 2].data_ptr(), src.data_ptr(), b.gbuf, level, 0, H)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


acc = {1: [0.0] * 5, 0: [0.0] * 5}
rounds = 4
for r in range(rounds):
    for fused in (1, 0):
        f.set_fused(fused)
        for level in range(5):
            acc[fused][level] += level_us(level) / rounds
print("level (step)      fused taps us   reference order us   ratio")
for level in range(5):
    print("  %d (%2d)          %9.1f        %9.1f          %.3f" % (level, 1 << level, acc[1][level], acc[0][level], acc[0][level] / acc[1][level]))
print("  all five        %9.1f        %9.1f          %.3f" % (sum(acc[1]), sum(acc[0]), sum(acc[0]) / sum(acc[1])))
