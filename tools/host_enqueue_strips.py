#!/usr/bin/env python3
"""tools/host_enqueue_strips.py -- host time to enqueue one frame of the C-ABI strip driver (rs_strips_frame + tone map +
rs_gbuffer_update + rs_strips_gather_begin / _end) for a middle strip of an N-way split, against the GPU time of that frame.
The transport is a stream-ordered no-op (callbacks that return at once): every host call of a real frame is made -- packing and
unpacking copies, events, the group calls -- except RCCL's own ncclSend / ncclRecv / ncclGroupEnd, and nothing travels, so the
image is wrong and only the TIMES mean something.  Shows whether a 1/8 strip (0.18 ms of kernels) is bound by the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from restir_amd import capi, scenes
from restir_amd.tiling import strip_bounds

W, H = 1920, 1080
capi.init(0)
sd = scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
capi.set_sync(False)
for world, rank in ((8, 3), (4, 1), (2, 0)):
    comm = capi.Comm(rank, world, lambda p, n, peer: None, lambda p, n, peer: None, None, None, stream_ordered=True)
    drv = capi.Strips(comm, W, H)
    gbuf, restir = capi.GBuffer(W, H), capi.ReSTIR(W, H)
    image = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda")
    pbos = [torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    y0, y1 = drv.y0, drv.y1
    state = {"n": 0}

    def frame():
        k = state["n"] % 2; state["n"] += 1
        drv.frame(restir, scene, cam, gbuf, image.data_ptr(), 0, state["n"], 3)
        gbuf.update(cam)
        drv.gather_end(k)
        capi.copy_image_to_pbo(pbos[k].data_ptr() + y0 * W * 4, image.data_ptr() + y0 * W * 12, W, y1 - y0, 2, 1.0)
        drv.gather_begin(pbos[k].data_ptr(), 4, 0, k)

    for _ in range(20):
        frame()
    capi.synchronize(); torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        frame()
    t1 = time.perf_counter()
    capi.synchronize(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("world %d rank %d (rows %d): host enqueue %.3f ms/frame, total %.3f ms/frame" % (world, rank, y1 - y0, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
    drv.gather_end(0); drv.gather_end(1)
    drv.destroy(); comm.destroy()
