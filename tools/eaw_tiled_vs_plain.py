#!/usr/bin/env python3
"""tools/eaw_tiled_vs_plain.py OUT.npy W H -- renders 3 frames of the Sponza-class scene at W x H, runs LeveledEAWFilter on the
radiance image and saves the filtered image.  Run once with EAW_TILED=1 and once with EAW_TILED=0 (handed to rs_eaw_set_tiled)
and compare the two files bit for bit: the LDS-tiled levels (steps 1, 2, 4) must equal the plain gathers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from restir_amd import capi, scenes
out, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
capi.init(0)
sd = scenes.sponza_class(seed=1, scale=0.1)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
gbuf, restir, eaw = capi.GBuffer(W, H), capi.ReSTIR(W, H), capi.EAWFilter(W, H, 5)
eaw.set_tiled(os.environ.get("EAW_TILED", "1") != "0")
image = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda")
result = torch.zeros_like(image)
for frame in range(3):
    gbuf.render(scene, cam)
    restir.direct(scene, cam, gbuf, image.data_ptr(), 0, frame, 3)
    ptr = eaw.filter(result.data_ptr(), image.data_ptr(), gbuf, cam)
    gbuf.update(cam)
capi.synchronize(); torch.cuda.synchronize()
got = torch.empty_like(image)
capi.hip_memcpy_d2d(got.data_ptr(), ptr, W * H * 12)
torch.cuda.synchronize()
np.save(out, got.cpu().numpy())
print("saved", out, "mean", float(got.mean()))
