#!/usr/bin/env python3
"""tools/bench_sync_frame.py [--bistro] -- one frame of the reference's runCuda() in the reference's own mode, a synchronisation after every
call (rs_set_sync(1), the library's default): GBuffer::render, ReSTIRDirect (rs_restir_direct), copyImageToPBO, GBuffer::update at 1080p.
TILE_SPLIT=<threshold> in the environment is handed to rs_set_tile_split (0 = off; default 768)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from restir_amd import capi, scenes

W, H = 1920, 1080
capi.init(0)
sd = scenes.bistro_class(seed=2, scale=1.0) if "--bistro" in sys.argv else scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
gbuf = capi.GBuffer(W, H); restir = capi.ReSTIR(W, H)
image = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda")
pbo = torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda")
capi.set_sync(True)
if "TILE_SPLIT" in os.environ:
    capi.set_tile_split(int(os.environ["TILE_SPLIT"]))


def frame(f):
    gbuf.render(scene, cam)
    restir.direct(scene, cam, gbuf, image.data_ptr(), 0, f, 3)
    capi.copy_image_to_pbo(pbo.data_ptr(), image.data_ptr(), W, H, 2, 1.0)
    gbuf.update(cam)


for f in range(20):
    frame(f)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 50
for f in range(20, 20 + N):
    frame(f)
torch.cuda.synchronize()
print("synchronous frame (render + ReSTIRDirect + tone map + update): %.3f ms; tile split %s" %
      ((time.perf_counter() - t0) / N * 1e3, os.environ.get("TILE_SPLIT", "default")))
