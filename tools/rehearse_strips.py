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
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
