#!/usr/bin/env python3
"""tools/strip_passes.py y0 y1 [y0 y1 ...] -- per-pass GPU time of single strips (finds rows whose waves set a kernel's tail)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from restir_amd import capi, scenes
from restir_amd.tiling import HipBackend

W, H = int(os.environ.get("RS_W", 1920)), int(os.environ.get("RS_H", 1080))
capi.init(0)
sd = scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
b = HipBackend(capi, scene, cam, W, H)
b.restir.enable_timing(True)
capi.set_sync(False)
capi.set_side_stream(0)          # per-pass times: one kernel at a time on the library stream
args = [int(a) for a in sys.argv[1:]]
for y0, y1 in zip(args[::2], args[1::2]):
    acc = [0.0] * 5
    n = 10
    for i in range(n + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.gbuffer_render(max(0, y0 - 5), min(H, y1 + 5)); e1.record()
        b.phase_a(i, 3, y0, y1); b.phase_b(0, 3, y0, y1); b.end_frame()
        torch.cuda.synchronize()
        if i >= 3:
            ms = b.restir.pass_times()
            acc[0] += e0.elapsed_time(e1)
            for k in range(4): acc[k + 1] += ms[k]
    print("rows %4d-%4d: gbuffer %.3f primary %.3f ris %.3f shadow %.3f spatial %.3f ms" % ((y0, y1) + tuple(a / n for a in acc)))
