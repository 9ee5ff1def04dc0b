#!/usr/bin/env python3
"""tools/strip_balance.py -- on ONE GPU: GPU time per frame of every strip of an N-way row decomposition
(no exchange), i.e. what each rank of `bench.py --gpus N` would spend in kernels.  Shows load imbalance
and the per-strip fixed cost that bounds strong scaling."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from restir_amd import capi, scenes
from restir_amd.tiling import HipBackend, StripRenderer, rebalance_bounds, strip_bounds

W, H = int(os.environ.get("RS_W", 1920)), int(os.environ.get("RS_H", 1080))
capi.init(0)
sd = scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
backend = HipBackend(capi, scene, cam, W, H)
capi.set_sync(False)
pbo = torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda")


def time_strip(world, rank, frames=30, bounds=None):
    s = StripRenderer(backend, world, rank, H, bounds=bounds)
    s.start_halo_exchange = lambda: ([], [], [])      # no neighbours here: the strip's kernels only
    def frame():
        s.frame(3, 0)
        capi.copy_image_to_pbo(pbo.data_ptr(), backend.image.data_ptr() + s.y0 * W * 12, W, s.y1 - s.y0, 2, 1.0)
    for _ in range(5):
        frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        frame()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3


t1 = time_strip(1, 0)
print("N=1: %.3f ms" % t1)
for n in (2, 4, 8):
    ts = [time_strip(n, r) for r in range(n)]
    print("N=%d: per-strip ms %s  max %.3f  ->  kernel-only speed-up %.2fx (ideal %d)" % (n, " ".join("%.3f" % t for t in ts), max(ts), t1 / max(ts), n))
    bounds = [strip_bounds(H, n, r) for r in range(n)]
    for it in range(4):
        bounds = rebalance_bounds(bounds, ts, H)
        ts = [time_strip(n, r, bounds=bounds) for r in range(n)]
        print("   balanced %d: rows %s  ms %s  max %.3f -> %.2fx" % (it + 1, " ".join(str(b - a) for a, b in bounds), " ".join("%.3f" % t for t in ts), max(ts), t1 / max(ts)))
