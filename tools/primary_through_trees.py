import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from restir_amd import capi, scenes
W, H = 1920, 1080
capi.init(0)
sd = scenes.bistro_class(seed=2, scale=1.0) if "--bistro" in sys.argv else scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
ys, xs = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
pos = torch.tensor([cam.position[0], cam.position[1], cam.position[2]], device="cuda")
fwd = torch.tensor([cam.view[0], cam.view[1], cam.view[2]], device="cuda"); right = torch.tensor([cam.right[0], cam.right[1], cam.right[2]], device="cuda"); up = torch.tensor([cam.up[0], cam.up[1], cam.up[2]], device="cuda")
tany = float(np.tan(np.radians(cam.fov[1]))); aspect = W / H
ndx = (1.0 - 2.0 * (xs.float() + 0.5) / W) * tany * aspect; ndy = (1.0 - 2.0 * (ys.float() + 0.5) / H) * tany
d = fwd[None, None, :] + right[None, None, :] * ndx[..., None] + up[None, None, :] * ndy[..., None]
d = d / d.norm(dim=-1, keepdim=True)
rays = torch.cat([pos.expand(H, W, 3), d], -1)                       # H x W x 6
# 8x8 tile order: 64 consecutive rays = one tile, as the packet kernels map them
tiles = rays.reshape(H // 8, 8, W // 8, 8, 6).permute(0, 2, 1, 3, 4).reshape(-1, 6).contiguous()
scan = rays.reshape(-1, 6).contiguous()
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for rep in range(2):
    for name, r in (("8x8 tiles", tiles), ("scan lines", scan)):
        print("%-10s rs_trace_closest (the reference tree): %.3f ms" % (name, timed(lambda: capi.trace_closest(scene, r))))
        for on in (True, False):
            capi.set_ordered_tree(scene, on)
            print("%-10s rs_trace_closest_wave, %s: %.3f ms" % (name, "closest-hit trees" if on else "reference tree, pair-cooperative", timed(lambda: capi.trace_closest_wave(scene, r))))
        capi.set_ordered_tree(scene, True)
# for scale: the product's kernels on the same rays
gbuf = capi.GBuffer(W, H)
print("k_render_gbuffer (packet walk, synchronous): %.3f ms" % timed(lambda: gbuf.render(scene, cam)))
