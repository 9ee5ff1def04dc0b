#!/usr/bin/env python3
"""tools/soak_glibc_full_size.py [frames] [bistro] -- BASELINE config 3 (or the Bistro-class scene) at 1920x1080 against the oracle in its PINNED
libm mode (glibc cosf / sinf in the spatial taps' disk mapping, src/restir.cu:47-56): `frames` spatiotemporal frames of an orbiting camera,
per frame the number of pixels whose radiance differs in any bit, the mean per-pixel L1 and the fraction beyond 1e-3 -- how often the one-ulp
difference between glibc's and the correctly rounded cos / sin moves a tap to the neighbouring pixel.  (Everything the taps do not feed is
asserted bit-exact by tests/test_gpu_full_size.py.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from restir_amd import capi
from restir_amd.scenes import orbit_position
from tests.common import HipRenderer, OracleRenderer, get_scene

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
name = "bistro:1.0" if "bistro" in sys.argv else "sponza:1.0"
W, H = 1920, 1080
capi.init(0)
sd = get_scene(name)
ob.set_libm_mode(0)
o = OracleRenderer(sd, W, H); h = HipRenderer(capi, sd, W, H)
total, worst, l1sum, flipsum = 0, 0, 0.0, 0.0
t0 = time.time()
for f in range(frames):
    p = orbit_position(sd.camera_args["position"], f, radius=1.0)
    o.set_camera_position(p); h.set_camera_position(p)
    a = o.frame(3); b = h.frame(3)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).sum(axis=1)
    n = int(np.count_nonzero((a.view(np.uint32) != b.view(np.uint32)).any(axis=1)))
    total += n; worst = max(worst, n); l1sum += float(d.mean()); flipsum += float(np.mean(d > 1e-3))
    if f % 10 == 9 or f == frames - 1:
        print("... frame %d: %d differing pixels so far (worst frame %d), %.0f s" % (f + 1, total, worst, time.time() - t0), flush=True)
print("%s, %dx%d, %d orbit frames against the oracle's glibc mode: %d differing pixels in %d (%.3g per frame, worst frame %d); mean per-pixel L1 %.3g "
      "(tolerance 1e-4), fraction beyond 1e-3 %.3g (tolerance 1e-3)" % (name, W, H, frames, total, frames * W * H, total / frames, worst, l1sum / frames, flipsum / frames))
