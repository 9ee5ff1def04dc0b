#!/usr/bin/env python3
"""tools/config5_single_gpu.py -- BASELINE config 5 (Bistro-class scene, 10 240 emissive triangles, 1080p spatiotemporal
ReSTIR-DI + the 5-level EAW denoiser) on ONE GPU: scene build time, ms/frame and per-pass times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from restir_amd import capi, scenes
from restir_amd.tiling import HipBackend
capi.init(0)
t=time.time(); sd = scenes.bistro_class(2, 1.0); print("gen %.1fs" % (time.time()-t), sd.num_prims, flush=True)
t=time.time(); scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials); print("build %.1fs" % (time.time()-t), flush=True)
W,H=1920,1080
cam = capi.camera_update(sd.camera(W,H))
b = HipBackend(capi, scene, cam, W, H)
eaw = capi.EAWFilter(W,H,5); out = torch.zeros_like(b.image)
capi.set_sync(False)
for f in range(40):                      # warm-up, and past the frames in which ReSTIRDirect measures its launch choice
    b.gbuffer_render(0,H); b.phase_a(f,3,0,H); b.phase_b(0,3,0,H); b.end_frame()
torch.cuda.synchronize()
t=time.time()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20          # timed frames
for f in range(40,40+N):
    b.gbuffer_render(0,H); b.phase_a(f,3,0,H); b.phase_b(0,3,0,H); p = eaw.filter(out.data_ptr(), b.image.data_ptr(), b.gbuf, cam); b.end_frame()
torch.cuda.synchronize(); dt=(time.time()-t)/N
# per-pass times: one kernel at a time (timing on keeps the primary-ray + RIS kernels on the library stream)
b.restir.enable_timing(True); capi.set_side_stream(0)
for f in range(40+N,45+N):
    b.gbuffer_render(0,H); b.phase_a(f,3,0,H); b.phase_b(0,3,0,H); b.end_frame()
torch.cuda.synchronize()
passes = b.restir.pass_times()
# the reference's mode: a synchronisation after every call
b.restir.enable_timing(False); capi.set_side_stream(4); capi.set_sync(True)
for f in range(45+N,50+N):
    b.gbuffer_render(0,H); b.phase_a(f,3,0,H); b.phase_b(0,3,0,H); eaw.filter(out.data_ptr(), b.image.data_ptr(), b.gbuf, cam); b.end_frame()
torch.cuda.synchronize(); t=time.time()
for f in range(50+N,60+N):
    b.gbuffer_render(0,H); b.phase_a(f,3,0,H); b.phase_b(0,3,0,H); eaw.filter(out.data_ptr(), b.image.data_ptr(), b.gbuf, cam); b.end_frame()
torch.cuda.synchronize(); ds=(time.time()-t)/10
print("config 5 on one GPU: %.2f ms/frame incl. EAW (frames overlapped), %.2f ms synchronous; pass ms on one stream %s; finite %s" % (dt*1e3, ds*1e3, passes, bool(torch.isfinite(b.image).all())))
