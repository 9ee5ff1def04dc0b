#!/usr/bin/env python3
"""tools/rehearse_strips.py -- the multi-process row-strip path (StripRenderer + HipBackend + torch.distributed) on a
ONE-GPU box: every rank uses the same card, the process group is gloo instead of RCCL.  Rank 0 also renders the full
frame by itself and checks that the gathered strips equal it bit for bit (static and orbiting camera).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29541 tools/rehearse_strips.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from restir_amd import capi, scenes
from restir_amd.scenes import orbit_position
from restir_amd.tiling import HipBackend, StripRenderer

W, H, FRAMES = 480, 270, 4
capi.init(0)
sd = scenes.sponza_class(seed=1, scale=0.1)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
ok = True
for moving, overlapped, denoise in ((False, False, False), (True, False, False), (False, True, False), (True, True, False), (True, False, True), (True, True, True)):
    # denoise: LeveledEAWFilter on the strips (border rows of every level exchanged), compared with the full-frame filter
    # overlapped: asynchronous launches (rs_set_sync(0)) -- frames overlap on the auxiliary streams, as in bench.py
    capi.set_sync(not overlapped)
    cam = capi.camera_update(sd.camera(W, H))
    strips = StripRenderer(HipBackend(capi, scene, cam, W, H), world, rank, H, dist=dist, share_history=moving)
    full = StripRenderer(HipBackend(capi, scene, cam, W, H), 1, 0, H) if rank == 0 else None
    for frame in range(FRAMES):
        if moving:
            p = orbit_position(sd.camera_args["position"], frame, radius=0.5)
            for i in range(3):
                cam.position[i] = float(p[i])
            capi.camera_update(cam)
        strips.frame(3, 0, denoise=denoise)
        if full is not None:
            full.frame(3, 0, denoise=denoise)
    torch.cuda.synchronize()
    capi.set_sync(True)
    src = strips.filtered if denoise else strips.b.image
    mine = src[strips.y0 * W:strips.y1 * W].contiguous()
    pad = torch.zeros((strips.max_rows * W, 3), dtype=torch.float32, device="cuda"); pad[:mine.shape[0]] = mine
    out = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, out, dst=0)
    if rank == 0:
        got = torch.cat([out[r][:(b[1] - b[0]) * W] for r, b in enumerate(strips.bounds)]).cpu().numpy()
        ref = (full.filtered if denoise else full.b.image).cpu().numpy()
        same = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        print("world %d, %s camera, %s launches%s: strips == full frame: %s" %
              (world, "orbiting" if moving else "static", "overlapped" if overlapped else "synchronous", ", EAW filter" if denoise else "", same), flush=True)
        ok = ok and same

# ---- the C-ABI strip driver (rs_comm / rs_strips, include/restir_hip.h) with gloo under its transport callbacks -------------------
# What a C++ caller runs over RCCL (rs_comm_create_rccl); here send / recv stage through host memory and torch.distributed.
from restir_amd.rccl import GlooTransport              # send / recv staged through host memory and torch.distributed (what bench.py's rehearsal mode uses)

for moving, overlapped, denoise in ((False, False, False), (False, True, False), (True, False, False), (True, True, True)):
    # moving: orbiting camera + rs_strips_exchange_history; denoise: rs_strips_eaw_filter; the image is assembled by rs_strips_gather
    capi.set_sync(not overlapped)
    cam = capi.camera_update(sd.camera(W, H))
    comm = GlooTransport(capi, dist, torch).comm(rank, world)
    drv = capi.Strips(comm, W, H)
    gbuf, restir = capi.GBuffer(W, H), capi.ReSTIR(W, H)
    eaw = capi.EAWFilter(W, H, 5) if denoise else None
    image = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda")
    full = StripRenderer(HipBackend(capi, scene, cam, W, H), 1, 0, H) if rank == 0 else None
    ref_py = StripRenderer(HipBackend(capi, scene, cam, W, H), world, rank, H, dist=dist, share_history=moving)    # the Python form of the same schedule
    result_ptr = image.data_ptr()
    for frame in range(FRAMES):
        if moving:
            p = orbit_position(sd.camera_args["position"], frame, radius=0.5)
            for i in range(3):
                cam.position[i] = float(p[i])
            capi.camera_update(cam)
        drv.frame(restir, scene, cam, gbuf, image.data_ptr(), 0, frame, 3)
        if denoise:
            result_ptr = drv.eaw_filter(eaw, gbuf, cam, image.data_ptr())
        gbuf.update(cam)
        if moving:
            drv.exchange_history(restir, gbuf)
        ref_py.frame(3, 0, denoise=denoise)
        if full is not None:
            full.frame(3, 0, denoise=denoise)
    if moving:
        drv.gather(result_ptr, 12, 0)                                 # image assembly on rank 0
    else:                                                             # the form bench.py uses: begun behind the frame, ended before the buffer is read
        drv.gather_begin(result_ptr, 12, 0, 1)
        drv.gather_end(1)
    torch.cuda.synchronize(); capi.synchronize()
    capi.set_sync(True)
    result = torch.empty((W * H, 3), dtype=torch.float32, device="cuda")
    capi.hip_memcpy_d2d(result.data_ptr(), result_ptr, W * H * 12)
    torch.cuda.synchronize()
    py = ref_py.filtered if denoise else ref_py.b.image
    same_as_python = bool(torch.equal(result[drv.y0 * W:drv.y1 * W].view(torch.int32), py[drv.y0 * W:drv.y1 * W].view(torch.int32)))
    assert (drv.y0, drv.y1) == (ref_py.y0, ref_py.y1)
    flags = [None] * world
    dist.all_gather_object(flags, same_as_python)
    if rank == 0:
        ref = full.filtered if denoise else full.b.image
        same = np.array_equal(result.cpu().numpy().view(np.uint32), ref.cpu().numpy().view(np.uint32))
        print("world %d, C-ABI strip driver, %s camera, %s launches%s: gathered strips == full frame: %s, == tiling.py on every rank: %s" %
              (world, "orbiting" if moving else "static", "overlapped" if overlapped else "synchronous", ", EAW filter" if denoise else "", same, all(flags)), flush=True)
        ok = ok and same and all(flags)
    drv.destroy(); comm.destroy()
# ---- SpatioTemporalFilter on strips (rs_strips_svgf_filter / rs_strips_exchange_svgf_history) against the full-frame filter ---------
W2, H2 = 320, 40 * world + 8                                          # strips of at least 33 rows
for moving, overlapped in ((False, False), (True, False), (True, True)):
    capi.set_sync(not overlapped)
    cam = capi.camera_update(sd.camera(W2, H2))
    comm = GlooTransport(capi, dist, torch).comm(rank, world)
    drv = capi.Strips(comm, W2, H2)
    gbuf, restir, svgf = capi.GBuffer(W2, H2), capi.ReSTIR(W2, H2), capi.SVGFFilter(W2, H2, 5)
    image = torch.zeros((W2 * H2, 3), dtype=torch.float32, device="cuda")
    fgb, frs, fsv = (capi.GBuffer(W2, H2), capi.ReSTIR(W2, H2), capi.SVGFFilter(W2, H2, 5)) if rank == 0 else (None, None, None)
    fimage = torch.zeros_like(image) if rank == 0 else None
    same = True
    for frame in range(5):
        if moving:
            p = orbit_position(sd.camera_args["position"], frame, radius=0.4)
            for i in range(3):
                cam.position[i] = float(p[i])
            capi.camera_update(cam)
        drv.frame(restir, scene, cam, gbuf, image.data_ptr(), 0, frame, 1)
        res_ptr = drv.svgf_filter(svgf, gbuf, cam, image.data_ptr())
        gbuf.update(cam)
        if moving:
            drv.exchange_history(restir, gbuf)
            drv.exchange_svgf_history(svgf)
        svgf.next_frame()
        drv.gather(res_ptr, 12, 0)
        capi.synchronize(); torch.cuda.synchronize()
        if rank == 0:
            fgb.render(scene, cam); frs.direct(scene, cam, fgb, fimage.data_ptr(), 0, frame, 1)
            ref_ptr = fsv.filter(fimage.data_ptr(), fgb, cam)
            fgb.update(cam); fsv.next_frame()
            capi.synchronize()
            a = torch.empty((W2 * H2, 3), dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
            capi.hip_memcpy_d2d(a.data_ptr(), res_ptr, W2 * H2 * 12); capi.hip_memcpy_d2d(b.data_ptr(), ref_ptr, W2 * H2 * 12)
            torch.cuda.synchronize()
            same = same and bool(torch.equal(a.view(torch.int32), b.view(torch.int32)))
    capi.set_sync(True)
    if rank == 0:
        print("world %d, C-ABI strip driver, SVGF filter, %s camera, %s launches: gathered strips == full-frame filter over 5 frames: %s" %
              (world, "orbiting" if moving else "static", "overlapped" if overlapped else "synchronous", same), flush=True)
        ok = ok and same
    drv.destroy(); comm.destroy()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
