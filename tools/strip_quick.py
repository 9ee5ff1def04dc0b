#!/usr/bin/env python3
"""tools/strip_quick.py -- on ONE GPU: ms per frame of the full frame and of the strips of an 8-way split with the balanced heights of
tools/strip_balance.py (no exchange): the quick form for A/B runs (environment variables select the library's variants)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from restir_amd import capi, scenes
from restir_amd.tiling import HipBackend, StripRenderer

W, H = 1920, 1080
capi.init(0)
sd = scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
backend = HipBackend(capi, scene, cam, W, H)
capi.set_sync(False)
if os.environ.get("RS_SS"):
    capi.set_side_stream(int(os.environ["RS_SS"]))
pbo = torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda")
ROWS8 = [152, 128, 120, 80, 88, 104, 176, 232]


def time_strip(world, rank, frames=40, bounds=None):
    s = StripRenderer(backend, world, rank, H, bounds=bounds)
    s.start_halo_exchange = lambda: ([], [], [])
    def frame():
        s.frame(3, 0)
        capi.copy_image_to_pbo(pbo.data_ptr(), backend.image.data_ptr() + s.y0 * W * 12, W, s.y1 - s.y0, 2, 1.0)
    for _ in range(20):
        frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        frame()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3

t1 = time_strip(1, 0)
b, y = [], 0
for r in ROWS8:
    b.append((y, y + r)); y += r
ts = [time_strip(8, r, bounds=b) for r in range(8)]
print("full %.3f ms; 8 strips %s  max %.3f -> %.2fx" % (t1, " ".join("%.3f" % t for t in ts), max(ts), t1 / max(ts)))
