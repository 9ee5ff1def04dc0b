#!/usr/bin/env python3
"""tools/bench_closest_wave.py [--bistro] -- the wave-level closest-hit service of the bounce rays (rs_trace_closest_wave) alone: 2 M first-bounce
rays of the 1080p bench view (origin = the camera ray's hit point, direction uniform over the hemisphere), in pixel order, through the
closest-hit trees in the reference's orders and through the reference's own tree, interleaved; next to it the shadow-ray service on
segments from the same origins to the hit points of other pixels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from restir_amd import capi, scenes

W, H = 1920, 1080
capi.init(0)
sd = scenes.bistro_class(seed=2, scale=1.0) if "--bistro" in sys.argv else scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
gbuf = capi.GBuffer(W, H)
gbuf.render(scene, cam)
view = gbuf.view()
torch.cuda.synchronize()
g = torch.Generator(device="cuda"); g.manual_seed(1)
# camera rays through the pixel centres -> hit points via the G-buffer's depth is not a position; trace them instead
ys, xs = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
pos = torch.tensor([cam.position[0], cam.position[1], cam.position[2]], device="cuda")
fwd = torch.tensor([cam.view[0], cam.view[1], cam.view[2]], device="cuda"); right = torch.tensor([cam.right[0], cam.right[1], cam.right[2]], device="cuda"); up = torch.tensor([cam.up[0], cam.up[1], cam.up[2]], device="cuda")
tany = float(np.tan(np.radians(cam.fov[1]))); aspect = W / H
ndx = (1.0 - 2.0 * (xs.float() + 0.5) / W) * tany * aspect; ndy = (1.0 - 2.0 * (ys.float() + 0.5) / H) * tany
d = fwd[None, None, :] + right[None, None, :] * ndx[..., None] + up[None, None, :] * ndy[..., None]
d = d / d.norm(dim=-1, keepdim=True)
rays = torch.cat([pos.expand(H, W, 3), d], -1).reshape(-1, 6).contiguous()
prim, mat, hp, hn = capi.trace_closest(scene, rays)
ok = prim >= 0
dd = torch.randn((W * H, 3), device="cuda", generator=g); dd = dd / dd.norm(dim=-1, keepdim=True)
dd = dd * torch.sign((dd * hn).sum(-1, keepdim=True))
b = torch.cat([hp + dd * 1e-5, dd], -1)[ok].contiguous()
perm = torch.randperm(int(ok.sum()), device="cuda", generator=g)
seg = torch.cat([hp[ok] + hn[ok] * 1e-4, hp[ok][perm]], -1).contiguous()
print("%d bounce rays, %d segments" % (b.shape[0], seg.shape[0]))


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for rep in range(2):
    for on in (True, False):
        capi.set_ordered_tree(scene, on)
        ms = timed(lambda: capi.trace_closest_wave(scene, b))
        print("closest hit, %-45s %.3f ms = %.3f ms per million rays" % ("trees in the reference's orders:" if on else "the reference's tree (pair-cooperative):", ms, ms / b.shape[0] * 1e6))
    ms = timed(lambda: capi.trace_occlusion(scene, seg))
    print("shadow segments (second tree):                            %.3f ms = %.3f ms per million" % (ms, ms / seg.shape[0] * 1e6))
capi.set_ordered_tree(scene, True)

if os.environ.get("WALK_STATS"):        # a -DRS_WALK_STATS build (tools/build_variant.sh stats -DRS_WALK_STATS; RESTIR_HIP_LIB=restir_amd/librestir_stats.so)
    import ctypes as C
    L = capi.lib(); out = (C.c_ulonglong * 32)()
    L.rs_debug_walk_stats_ordered.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert L.rs_debug_walk_stats_ordered(scene.handle, out, 1) == 0
    capi.trace_closest_wave(scene, b)
    assert L.rs_debug_walk_stats_ordered(scene.handle, out, 1) == 0
    w = out[0]
    print("closest-hit trees, per wave (%d waves): walk iterations %.1f with %.1f lanes walking; leaf rounds %.1f; triangle iterations %.1f with %.1f lanes; verify iterations %.1f with %.1f lanes"
          % (w, out[1] / w, out[2] / max(out[1], 1), out[3] / w, out[4] / w, out[5] / max(out[4], 1), out[6] / w, out[7] / max(out[6], 1)))
    print("per ray: %.1f node steps" % (out[9] / max(out[8], 1)))
    capi.set_ordered_tree(scene, False)
    capi.trace_closest_wave(scene, b)
    assert L.rs_debug_walk_stats_ordered(scene.handle, out, 1) == 0
    capi.set_ordered_tree(scene, True)
    print("the reference's tree, per wave (%d waves): iterations %.1f with %.1f lanes walking = %.1f node steps per ray" % (out[16], out[17] / out[16], out[18] / max(out[17], 1), out[18] / b.shape[0]))
