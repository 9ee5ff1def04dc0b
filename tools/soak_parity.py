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
names = ["cornell", "sponza:0.05", "bistro:0.03", "cornell_textured"]
scenes = {n: get_scene(n) for n in names}
osc = {n: oracle_scene(s) for n, s in scenes.items()}
hsc = {n: hip_scene(capi, s) for n, s in scenes.items()}
t0 = time.time(); runs = 0; px = 0; bad = 0; worst = 0.0
W, H = 192, 108
while time.time() - t0 < budget:
    n = names[runs % len(names)]
    sd = scenes[n]
    ob.set_libm_mode(1 if n == "cornell_textured" else mode)
    o = OracleRenderer(sd, W, H, scene=osc[n]); h = HipRenderer(capi, sd, W, H, scene=hsc[n])
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
    o.looper = h.looper = int(rng.integers(0, 1 << 20))
    for frame in range(3):
        a = o.frame(3); b = h.frame(3)
        ne = (a.view(np.uint32) != b.view(np.uint32)).any(axis=1)
        px += len(ne); bad += int(ne.sum())
        if ne.any():
            worst = max(worst, float(np.abs(a - b).sum(1).max()))
            print("mismatch: scene %s run %d frame %d pixels %d maxL1 %.3g" % (n, runs, frame, int(ne.sum()), worst), flush=True)
    runs += 1
ob.set_libm_mode(0)
print("libm mode %d;" % mode, "soak: %d runs, %d pixel-frames, %d with different bits (worst L1 %.3g) in %.0f s" % (runs, px, bad, worst, time.time() - t0))
