#!/usr/bin/env python3
"""tools/soak_parity.py [minutes] [libm mode] -- randomized parity soak: random cameras on several scenes, spatiotemporal ReSTIR-DI
for a few frames each, librestir_hip against the CPU oracle; counts pixels whose radiance bits differ.  libm mode 1
(default) = the oracle rounds its sin / cos / atan2 correctly, as the device does; 0 = glibc's float functions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from restir_amd import capi
from tests.common import HipRenderer, OracleRenderer, get_scene, hip_scene, oracle_scene

capi.init(0)
budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 180
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(int(time.time()))
names = os.environ.get("RS_SCENES", "cornell,sponza:0.05,bistro:0.03,cornell_textured").split(",")
from restir_amd import sobol
table = sobol.sobol_table()                            # every second pass over the scenes runs the Sobol sampler branch (src/sampler.h:9-36)
scenes = {n: get_scene(n) for n in names}
osc = {n: oracle_scene(s) for n, s in scenes.items()}
hsc = {n: hip_scene(capi, s) for n, s in scenes.items()}
last_note = time.time()
t0 = time.time(); runs = 0; px = 0; bad = 0; worst = 0.0; gi_px = 0; gi_bad = 0
W, H = int(os.environ.get("RS_W", 192)), int(os.environ.get("RS_H", 108))
while time.time() - t0 < budget:
    n = names[runs % len(names)]
    sd = scenes[n]
    ob.set_libm_mode(1 if n == "cornell_textured" else mode)
    use_sobol = (runs // len(names)) % 2 == 1
    o = OracleRenderer(sd, W, H, scene=osc[n], sobol=table if use_sobol else None); h = HipRenderer(capi, sd, W, H, scene=hsc[n], sobol=table if use_sobol else None)
    if not use_sobol:
        osc[n].set_sample_sequence(None); hsc[n].set_sample_sequence(None)
    base = np.array(sd.camera_args["position"], np.float64)
    for r in (o, h):
        r.cam.rotation[0] = sd.camera_args["rotation"][0] + 0.0
    jitter = rng.normal(size=3) * 0.4
    yaw, pitch = rng.uniform(-25, 25), rng.uniform(-10, 10)
    for r, upd in ((o, ob.camera_update), (h, capi.camera_update)):
        r.cam.rotation[0] = float(sd.camera_args["rotation"][0] + yaw); r.cam.rotation[1] = float(sd.camera_args["rotation"][1] + pitch)
        for i in range(3):
            r.cam.position[i] = float(base[i] + jitter[i])
        upd(r.cam)
    o.looper = h.looper = int(rng.integers(0, 10000 if use_sobol else 1 << 20))
    for frame in range(3):
        a = o.frame(3); b = h.frame(3)
        ne = (a.view(np.uint32) != b.view(np.uint32)).any(axis=1)
        px += len(ne); bad += int(ne.sum())
        if ne.any():
            worst = max(worst, float(np.abs(a - b).sum(1).max()))
            print("mismatch: scene %s run %d frame %d pixels %d maxL1 %.3g" % (n, runs, frame, int(ne.sum()), worst), flush=True)
    if mode == 1:                                   # multi-bounce kernels (their BSDF sampling needs the correctly rounded mode)
        import torch
        d0 = np.zeros((W * H, 3), np.float32); i0 = np.zeros_like(d0); i1 = np.zeros_like(d0)
        d1 = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda"); j0 = torch.zeros_like(d1); j1 = torch.zeros_like(d1)
        depth = int(rng.integers(1, 6)); lp = int(rng.integers(0, 9999 if use_sobol else 1 << 20))
        ra = ob.path_trace(o.scene, o.cam, d0, i0, 0, lp, depth); rb = capi.path_trace(h.scene, h.cam, d1.data_ptr(), j0.data_ptr(), 0, lp, depth)
        o.gbuf.render(o.scene, o.cam); h.gbuf.render(h.scene, h.cam)
        rc = o.restir.indirect(o.scene, o.cam, o.gbuf, i1, 0, lp + 1, 1, depth); rd = h.restir.indirect(h.scene, h.cam, h.gbuf, j1.data_ptr(), 0, lp + 1, 1, depth)
        for name_, a_, b_ in (("pathTrace direct", d0, d1), ("pathTrace indirect", i0, j0), ("ReSTIRIndirect", i1, j1)):
            ne = (a_.view(np.uint32) != b_.cpu().numpy().view(np.uint32)).any(axis=1)
            gi_px += len(ne); gi_bad += int(ne.sum())
            if ne.any():
                print("mismatch: scene %s run %d %s pixels %d" % (n, runs, name_, int(ne.sum())), flush=True)
        if ra != rb or rc != rd:
            print("ray-count mismatch: scene %s run %d" % (n, runs), flush=True); gi_bad += 1
    runs += 1
    if time.time() - last_note > 45:
        last_note = time.time()
        print("... %d runs, %d pixel-frames, %d differing" % (runs, px, bad + gi_bad), flush=True)
ob.set_libm_mode(0)
print("multi-bounce kernels: %d pixel-images, %d with different bits" % (gi_px, gi_bad))
print("scenes %s at %dx%d, default engine and Sobol sampler in turn;" % (",".join(names), W, H))
print("libm mode %d;" % mode, "soak: %d runs, %d pixel-frames, %d with different bits (worst L1 %.3g) in %.0f s" % (runs, px, bad, worst, time.time() - t0))
