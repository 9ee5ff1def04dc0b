#!/usr/bin/env python3
"""tools/ab_eaw_fused.py -- LeveledEAWFilter with its taps in the reference's separately rounded operations against the fused form
(rs_eaw_set_fused): interleaved timings at 1080p on the bench scene, the largest relative difference between the two results, and the
error of each against the oracle on the small scene of tests/test_gpu_parity.py::test_eaw_filter (rtol 1e-5 is the filter's stated
tolerance).  Run on the GPU box from the repo root."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from restir_amd import capi, scenes
from restir_amd.tiling import HipBackend

capi.init(0)


def filtered(f, image, gbuf, cam):
    out = torch.zeros_like(image)
    p = f.filter(out.data_ptr(), image.data_ptr(), gbuf, cam)
    capi.synchronize()
    res = torch.empty_like(image)
    capi.hip_memcpy_d2d(res.data_ptr(), p, res.numel() * 4)
    return res.cpu().numpy()


def rel(a, b):
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-6)).max())


# ---- against the oracle, small scene -------------------------------------------------------------------------------------------
from common import get_scene, OracleRenderer, HipRenderer
from oracle import binding as ob
ob.set_libm_mode(1)
sd = get_scene("sponza:0.03")
W, H = 160, 96
o = OracleRenderer(sd, W, H); h = HipRenderer(capi, sd, W, H)
o.gbuf.render(o.scene, o.cam); h.gbuf.render(h.scene, h.cam)
o.restir.direct(o.scene, o.cam, o.gbuf, o.image, 0, 0, 3)
h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, 0, 3)
ref = ob.eaw_filter(o.gbuf, o.cam, o.image)
f = capi.EAWFilter(W, H, 5)
for fused in (0, 1):
    f.set_fused(fused)
    got = filtered(f, h.image, h.gbuf, h.cam)
    print("fused %d: against the oracle (160x96, 5 levels) max relative error %.3g, max absolute %.3g, allclose(rtol 1e-5, atol 1e-6) %s"
          % (fused, rel(got, ref), float(np.abs(got - ref).max()), bool(np.allclose(ref, got, rtol=1e-5, atol=1e-6))))
f.destroy()

# ---- timings, 1080p ------------------------------------------------------------------------------------------------------------
W, H = 1920, 1080
sd = scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
b = HipBackend(capi, scene, cam, W, H)
for i in range(3):
    b.gbuffer_render(0, H); b.phase_a(i, 3, 0, H); b.phase_b(0, 3, 0, H); b.restir.end_frame()
torch.cuda.synchronize()
f = capi.EAWFilter(W, H, 5)
res = {}
for fused in (0, 1):
    f.set_fused(fused)
    res[fused] = filtered(f, b.image, b.gbuf, cam)
print("1080p: fused against separately rounded: max relative difference %.3g, pixels that differ %d of %d"
      % (rel(res[1], res[0]), int((res[1] != res[0]).any(axis=-1).sum()) if res[0].ndim > 1 else -1, W * H))
capi.set_sync(False)
out = torch.zeros_like(b.image)
state = {"p": out.data_ptr()}


def timed(reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        state["p"] = f.filter(state["p"], b.image.data_ptr(), b.gbuf, cam)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t = {0: [], 1: []}
for rnd in range(6):
    for fused in (0, 1):
        f.set_fused(fused)
        timed(3)
        t[fused].append(timed())
for fused in (0, 1):
    print("fused %d: %s us per call, median %.1f" % (fused, " ".join("%.1f" % v for v in t[fused]), float(np.median(t[fused]))))
f.destroy()

# ---- SpatioTemporalFilter: timings at 1080p, then the error against the oracle over 7 frames of an orbiting camera --------------------
svgf = capi.SVGFFilter(W, H, 5)


def svgf_timed(reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        svgf.filter(b.image.data_ptr(), b.gbuf, cam); svgf.next_frame()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t = {0: [], 1: []}
for rnd in range(6):
    for fused in (0, 1):
        svgf.set_fused(fused)
        svgf_timed(3)
        t[fused].append(svgf_timed())
for fused in (0, 1):
    print("SVGF fused %d: %s us per frame, median %.1f" % (fused, " ".join("%.1f" % v for v in t[fused]), float(np.median(t[fused]))))
svgf.destroy()
capi.set_sync(True)

from restir_amd.scenes import orbit_position
sd = get_scene("sponza:0.03")
W, H = 160, 96
for fused in (0, 1):
    o = OracleRenderer(sd, W, H); h = HipRenderer(capi, sd, W, H)
    fo = ob.SVGF(W, H); fh = capi.SVGFFilter(W, H, 5)
    fh.set_fused(fused)
    worst, worst_abs, ok = 0.0, 0.0, True
    for frame in range(7):
        p = orbit_position(sd.camera_args["position"], frame, radius=0.3)
        o.set_camera_position(p); h.set_camera_position(p)
        o.gbuf.render(o.scene, o.cam); h.gbuf.render(h.scene, h.cam)
        o.restir.direct(o.scene, o.cam, o.gbuf, o.image, 0, o.looper, 1)
        h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, h.looper, 1)
        o.looper += 1; h.looper += 1
        ref = fo.filter(o.image, o.gbuf, o.cam)
        ptr = fh.filter(h.image.data_ptr(), h.gbuf, h.cam)
        capi.synchronize()
        tt = torch.empty(W * H * 3, dtype=torch.float32, device="cuda")
        capi.hip_memcpy_d2d(tt.data_ptr(), ptr, W * H * 12)
        got = tt.cpu().numpy().reshape(-1, 3)
        worst = max(worst, rel(got, ref)); worst_abs = max(worst_abs, float(np.abs(got - ref).max()))
        ok = ok and bool(np.allclose(ref, got, rtol=3e-5, atol=2e-6))
        fo.next_frame(); fh.next_frame()
        o.gbuf.update(o.cam); h.gbuf.update(h.cam)
    print("SVGF fused %d: against the oracle over 7 frames max relative error %.3g, max absolute %.3g, allclose(rtol 3e-5, atol 2e-6) %s"
          % (fused, worst, worst_abs, ok))
    fh.destroy()
