#!/usr/bin/env python3
"""tools/probe_fused_walk.py -- measurement for a design decision: the G-buffer ray (pixel centre) and the shading ray (jittered)
of a pixel walked in ONE packet walk (restir_amd/csrc/probe_fused.hip) against two separate packet walks, 1080p bench scene.
Prints the kernel times and checks that both give the same primitives."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from restir_amd import capi, scenes
W, H = int(os.environ.get("RS_W", 1920)), int(os.environ.get("RS_H", 1080))
capi.init(0)
sd = scenes.sponza_class(seed=1, scale=1.0) if os.environ.get("RS_SCENE", "sponza") == "sponza" else scenes.bistro_class(2, 1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
L = capi.lib()
L.rs_debug_probe_fused.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
n = W * H
out = {m: [torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"),
           torch.zeros(n, dtype=torch.float32, device="cuda"), torch.zeros(n, dtype=torch.float32, device="cuda")] for m in (0, 1)}
capi.set_sync(False)
for m in (0, 1):
    ts = []
    for it in range(25):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        capi.check(L.rs_debug_probe_fused(scene.handle, C.byref(cam), m, it % 4, *[t.data_ptr() for t in out[m]]))
        e1.record(); torch.cuda.synchronize()
        if it >= 5: ts.append(e0.elapsed_time(e1))
    print("%s: %.3f ms (min %.3f)" % ("two separate packet walks" if m == 0 else "one fused packet walk   ", float(np.median(ts)), min(ts)))
same = all(torch.equal(out[0][k], out[1][k]) for k in (0, 1))
print("same primitives for both rays of every pixel:", same)
