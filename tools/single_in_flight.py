#!/usr/bin/env python3
"""tools/single_in_flight.py -- one runCuda frame at a time (the host waits for every frame: src/preview.cpp:337-361) with a synchronisation after
every call (rs_set_sync(1), the reference's mode) and with asynchronous launches (rs_set_sync(0)), configs 3, 4 and 5, interleaved, median of
13 frames each.  Asynchronous launches must never be the slower of the two."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from restir_amd import capi, scenes
from restir_amd.tiling import HipBackend
cfgs = {3: ("sponza", 1920, 1080, False), 4: ("sponza", 3840, 2160, False), 5: ("bistro", 1920, 1080, True)}
capi.init(0)
built = {}
for c in (3, 4, 5):
    name, W, H, den = cfgs[c]
    if name not in built:
        sd = scenes.sponza_class(seed=1, scale=1.0) if name == "sponza" else scenes.bistro_class(seed=2, scale=1.0)
        built[name] = (sd, capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials))
    sd, scene = built[name]
    cam = capi.camera_update(sd.camera(W, H))
    b = HipBackend(capi, scene, cam, W, H)
    eaw = capi.EAWFilter(W, H, 5) if den else None
    out = torch.zeros_like(b.image) if den else None
    pbo = torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda")
    capi.set_internal_stream_priority(1 if den else 2)
    st = {"n": 0}
    def frame():
        b.gbuffer_render(0, H); b.phase_a(st["n"], 3, 0, H); b.phase_b(0, 3, 0, H)
        shown = eaw.filter(out.data_ptr(), b.image.data_ptr(), b.gbuf, cam) if den else b.image.data_ptr()
        b.end_frame(); st["n"] += 1
        capi.copy_image_to_pbo(pbo.data_ptr(), shown, W, H, 2, 1.0)
    res = {}
    for mode in ("sync", "single", "sync", "single"):
        capi.set_sync(mode == "sync")
        ts = []
        for i in range(16):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            frame(); capi.synchronize(); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        res.setdefault(mode, []).append(float(np.median(ts[3:])))
    print("config %d: synchronous %s ms, one frame in flight %s ms" % (c, ["%.3f" % x for x in res["sync"]], ["%.3f" % x for x in res["single"]]), flush=True)
    capi.set_sync(True)
    if eaw: eaw.destroy()
