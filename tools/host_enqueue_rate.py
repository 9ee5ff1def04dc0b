#!/usr/bin/env python3
"""tools/host_enqueue_rate.py -- host time to enqueue a frame vs the frame time, for strips of an 8-way split and the full frame
(asynchronous mode).  Shows whether a small strip is bound by the host, by the GPU, or by cross-stream event latency
(try GPU_MAX_HW_QUEUES=2 / 8: more hardware queues made the overlapped frames SLOWER, 0.21 -> 0.33 ms on the lightest strip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from restir_amd import capi, scenes
from restir_amd.tiling import HipBackend, StripRenderer, strip_bounds
W, H = 1920, 1080
capi.init(0)
sd = scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
backend = HipBackend(capi, scene, cam, W, H)
capi.set_sync(False)
pbo = torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda")
for world, rank in ((8, 7), (8, 3), (8, 0), (1, 0)):
    s = StripRenderer(backend, world, rank, H)
    s.start_halo_exchange = lambda: ([], [], [])
    def frame():
        s.frame(3, 0)
        capi.copy_image_to_pbo(pbo.data_ptr(), backend.image.data_ptr() + s.y0 * W * 12, W, s.y1 - s.y0, 2, 1.0)
    for _ in range(5): frame()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): frame()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("world %d rank %d: host enqueue %.3f ms/frame, total %.3f ms/frame" % (world, rank, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
