#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- inputs and EXPECTED OUTPUTS only (no reference text).

Expected outputs come from
  * oracle/_ref/libref_subset.so : the reference's own functions compiled from /root/reference/src
    (intersectTriangle, AABB::intersect, BVHBuilder::build, Camera::*, Material::BSDF, Math::*, linearSample);
  * oracle/_ref/libthrust_probe.so : rocThrust's minstd_rand + uniform_real_distribution<float>.
Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
The GPU box has no /root/reference; tests/test_oracle_golden.py replays these vectors there and here.

A second file, frames_oracle.npz, holds frame-level outputs of the CPU oracle itself (NOT
reference-derived; a regression pin so that an accidental change of the oracle is noticed).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import binding as ob  # noqa: E402
from restir_amd.ctypes_structs import MATERIAL_DTYPE, make_camera  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def unit(rng, n):
    v = rng.normal(size=(n, 3)).astype(np.float32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def special_rays(rng, n):
    o = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    d = unit(rng, n)
    k = n // 8
    d[:k] = 0; d[np.arange(k), rng.integers(0, 3, k)] = rng.choice([-1.0, 1.0], k)      # axis aligned
    d[k:2 * k, 0] = rng.uniform(-1e-6, 1e-6, k)                                          # near-zero x
    d[2 * k:3 * k, 1] = 0.0                                                              # exact-zero y
    d[3 * k:4 * k, 2] = rng.uniform(-1e-7, 1e-7, k)                                      # near-zero z
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([o, d], 1), np.float32)


def main():
    R = ob.ref_subset(); T = ob.thrust_probe()
    assert R is not None and T is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20241003)
    g = {}

    n = 4000
    rays = special_rays(rng, n)
    tris = rng.uniform(-1.5, 1.5, (n, 9)).astype(np.float32)
    hit = np.zeros(n, np.int32); bary = np.zeros((n, 2), np.float32); dist = np.zeros(n, np.float32)
    R.ref_intersect_triangle(n, rays.reshape(-1), tris.reshape(-1), hit, bary.reshape(-1), dist)
    g.update(tri_rays=rays, tri_tris=tris, tri_hit=hit, tri_bary=bary, tri_dist=dist)

    boxes = np.sort(rng.uniform(-2, 2, (n, 2, 3)).astype(np.float32), axis=1).reshape(n, 6).copy()
    bh = np.zeros(n, np.int32); bt = np.zeros(n, np.float32)
    R.ref_aabb_intersect(n, rays.reshape(-1), boxes.reshape(-1), bh, bt)
    g.update(box_boxes=boxes, box_hit=bh, box_tmin=bt)

    x = rng.integers(0, 2 ** 32, 2000, dtype=np.uint32); h = np.zeros_like(x)
    R.ref_utilhash(len(x), x, h)
    g.update(hash_in=x, hash_out=h)

    seeds = rng.integers(-2 ** 31, 2 ** 31, 128).astype(np.int32); seeds[:4] = [0, 2147483647, -1, 1]
    stream = np.zeros((128, 200), np.float32)
    T.thr_rng_stream_raw(128, seeds, 200, stream.reshape(-1))
    g.update(rng_seeds=seeds, rng_stream=stream, thrust_version=np.int32(T.thr_version()))

    m = 3000
    mats = np.zeros(m, MATERIAL_DTYPE)
    mats["type"] = rng.integers(0, 5, m); mats["baseColor"] = rng.uniform(0, 1, (m, 3))
    mats["metallic"] = rng.uniform(0, 1, m); mats["roughness"] = rng.uniform(0.05, 1, m); mats["ior"] = 1.5
    for k in ("baseColorMapId", "metallicMapId", "roughnessMapId", "normalMapId"):
        mats[k] = -1
    nr, wo, wi = unit(rng, m), unit(rng, m), unit(rng, m)
    f = np.zeros((m, 3), np.float32)
    R.ref_bsdf(m, mats.ctypes.data, nr.reshape(-1), wo.reshape(-1), wi.reshape(-1), f.reshape(-1))
    g.update(bsdf_mats=mats.view(np.uint8).reshape(m, 44), bsdf_n=nr, bsdf_wo=wo, bsdf_wi=wi, bsdf_out=f)

    cams = []
    for i, (w, hgt, pos, rot, fov) in enumerate([(256, 256, (0, 1, 3.5), (-90, 0, 0), 19.5), (1920, 1080, (3, 2, -7), (37, -12, 0), 30.0)]):
        cam = make_camera(w, hgt, pos, rot, fov)
        R.ref_camera_update(C.byref(cam))
        k = 1500
        xy = np.stack([rng.integers(0, w, k), rng.integers(0, hgt, k)], 1).astype(np.int32)
        r4 = rng.uniform(0, 1, (k, 4)).astype(np.float32)
        cr = np.zeros((k, 6), np.float32)
        R.ref_camera_sample(C.byref(cam), k, xy.reshape(-1), r4.reshape(-1), cr.reshape(-1))
        dd = rng.uniform(0.1, 20, k).astype(np.float32); pp = np.zeros((k, 3), np.float32)
        R.ref_camera_position(C.byref(cam), k, xy.reshape(-1), dd, pp.reshape(-1))
        rc = np.zeros((k, 2), np.int32)
        R.ref_camera_raster_coord(C.byref(cam), k, pp.reshape(-1), rc.reshape(-1))
        g.update({f"cam{i}_args": np.array([w, hgt, *pos, *rot, fov], np.float64),
                  f"cam{i}_struct": np.frombuffer(bytes(cam), np.uint8).copy(),
                  f"cam{i}_xy": xy, f"cam{i}_r4": r4, f"cam{i}_rays": cr, f"cam{i}_dist": dd, f"cam{i}_pos": pp, f"cam{i}_raster": rc})

    ruv = rng.uniform(0, 1, (n, 2)).astype(np.float32); ruv[:5] = [[0, 0], [1, 1], [1, 0], [0, 1], [.5, .5]]
    st = np.zeros((n, 3), np.float32); R.ref_sample_triangle_uniform(n, tris.reshape(-1), ruv.reshape(-1), st.reshape(-1))
    dk = np.zeros((n, 2), np.float32); R.ref_to_concentric_disk(n, ruv.reshape(-1), dk.reshape(-1))
    area = np.zeros(n, np.float32); nrm = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
    pts = rays[:, :3].copy()
    R.ref_triangle_misc(n, tris.reshape(-1), pts.reshape(-1), area, nrm.reshape(-1), pdf)
    g.update(ruv=ruv, sample_tri=st, disk=dk, tri_area=area, tri_normal=nrm, tri_pdf=pdf, misc_x=pts)

    col = rng.uniform(0, 8, (2000, 3)).astype(np.float32)
    for mode in (0, 1, 2):
        o = np.zeros_like(col); R.ref_tonemap(len(col), col.reshape(-1), mode, o.reshape(-1)); g[f"tonemap{mode}"] = o
    g["tonemap_in"] = col

    for name, nt in (("a", 7), ("b", 400)):
        v = (rng.uniform(-5, 5, (nt, 1, 3)) + rng.uniform(-.3, .3, (nt, 3, 3))).astype(np.float32)
        if name == "b":
            v[:60, :, 1] = 0.25                                    # coplanar cluster: dimMax == dimMin nodes (Q12)
        bb, nn = ob.bvh_build(v, R.ref_bvh_build)
        g.update({f"bvh_{name}_verts": v, f"bvh_{name}_boxes": bb, f"bvh_{name}_nodes": nn})

    # image.h linearSample, mathUtil.h toSphere / toPlane / localToWorld (the texture and environment-map paths)
    tex = rng.uniform(0, 2, (13, 21, 3)).astype(np.float32)
    uv = rng.uniform(-3, 3, (3000, 2)).astype(np.float32)
    uv[:200] = rng.integers(-2, 3, (200, 2)).astype(np.float32)                     # texel borders / wrap
    uv[200:400] = (rng.integers(0, 21, (200, 2)) / np.float32(21)).astype(np.float32)
    ls = np.zeros((len(uv), 3), np.float32); R.ref_linear_sample(21, 13, tex.reshape(-1), len(uv), uv.reshape(-1), ls.reshape(-1))
    u01 = rng.uniform(0, 1, (3000, 2)).astype(np.float32)
    sph = np.zeros((len(u01), 3), np.float32); R.ref_to_sphere(len(u01), u01.reshape(-1), sph.reshape(-1))
    dirs = unit(rng, 3000); dirs[:50, 0] = 0; dirs[50:100, 2] = 0; dirs[100:150] = [0, 1, 0]; dirs[150:200] = [0, -1, 0]
    pl = np.zeros((len(dirs), 2), np.float32); R.ref_to_plane(len(dirs), dirs.reshape(-1), pl.reshape(-1))
    nn_ = unit(rng, 3000); nn_[:50] = [0, 1, 0]; nn_[50:100] = [0, -1, 0]
    lv = rng.uniform(-1, 1, (3000, 3)).astype(np.float32)
    l2w = np.zeros((3000, 3), np.float32); R.ref_local_to_world(3000, nn_.reshape(-1), lv.reshape(-1), l2w.reshape(-1))
    g.update(tex=tex, tex_uv=uv, tex_sample=ls, sph_uv=u01, sph_dir=sph, plane_dir=dirs, plane_uv=pl, l2w_n=nn_, l2w_v=lv, l2w_out=l2w)

    # Material::sample / pdf (material.h:230-256) for every type, poles and degenerate parameters included
    nm = 6000
    ms = np.zeros(nm, MATERIAL_DTYPE)
    ms["type"] = rng.integers(0, 5, nm); ms["baseColor"] = rng.uniform(0, 1, (nm, 3))
    ms["metallic"] = rng.uniform(0, 1, nm); ms["roughness"] = rng.uniform(0.02, 1, nm); ms["ior"] = rng.uniform(1.0, 2.5, nm)
    ms["metallic"][:100] = 0; ms["metallic"][100:200] = 1; ms["roughness"][200:300] = 0
    mn, mwo, mwi = unit(rng, nm), unit(rng, nm), unit(rng, nm)
    mn[:60] = [0, 1, 0]; mwo[60:120] = mn[60:120]
    mr = rng.uniform(0, 1, (nm, 3)).astype(np.float32); mr[:10] = 0; mr[10:20] = 1
    sd_, sb_, sp_, st_ = ob.material_sample(ms, mn, mwo, mr, R.ref_material_sample)
    g.update(ms_mats=ms, ms_n=mn, ms_wo=mwo, ms_wi=mwi, ms_r=mr, ms_dir=sd_, ms_bsdf=sb_, ms_pdf=sp_, ms_type=st_,
             ms_pdf_eval=ob.material_pdf(ms, mn, mwo, mwi, R.ref_material_pdf))

    np.savez_compressed(os.path.join(OUT, "functions_ref.npz"), **g)
    print("functions_ref.npz", os.path.getsize(os.path.join(OUT, "functions_ref.npz")), "bytes,", len(g), "arrays")

    # ---- frame-level regression pin of the oracle itself (NOT reference-derived) ----
    from tests.common import OracleRenderer, get_scene
    fr = {}
    sd = get_scene("cornell")
    for reuse in (0, 1, 2, 3):
        o = OracleRenderer(sd, 64, 64)
        for frame in range(3):
            img = o.frame(reuse)
        fr[f"cornell64_reuse{reuse}_frame2"] = img.copy()
        fr[f"cornell64_reuse{reuse}_M"] = o.restir.last["numSamples"].copy()
    o = OracleRenderer(sd, 64, 64)
    fr["cornell64_ptdirect"] = o.frame(0, use_reservoir=False).copy()
    o.gbuf.c.frameIdx ^= 1
    fr["cornell64_gbuf_id"] = o.gbuf.prim_id[o.gbuf.frame_idx].copy()
    fr["cornell64_gbuf_depth"] = o.gbuf.depth[o.gbuf.frame_idx].copy()
    for name in ("cornell_textured", "cornell_maps"):      # texture / environment-map paths, both libm modes
        for mode in (0, 1):
            ob.set_libm_mode(mode)
            o = OracleRenderer(get_scene(name), 64, 48)
            for frame in range(2):
                img = o.frame(3)
            fr[f"{name}_libm{mode}_frame1"] = img.copy()
            fr[f"{name}_libm{mode}_albedo"] = o.gbuf.albedo.copy()
    ob.set_libm_mode(0)
    # multi-bounce kernels (singleKernelPT, PTIndirectKernel, ReSTIRIndirectKernel)
    o = OracleRenderer(get_scene("cornell"), 48, 48)
    d = np.zeros((48 * 48, 3), np.float32); i1 = np.zeros_like(d); i2 = np.zeros_like(d); i3 = np.zeros_like(d)
    ob.path_trace(o.scene, o.cam, d, i1, 0, 0, 4)
    ob.pt_indirect(o.scene, o.cam, i2, 0, 1, 4)
    for frame in range(3):
        o.gbuf.render(o.scene, o.cam); o.restir.indirect(o.scene, o.cam, o.gbuf, i3, 0, frame, 1, 4); o.gbuf.update(o.cam)
    fr.update(cornell48_pt_direct=d, cornell48_pt_indirect=i1, cornell48_ptind=i2, cornell48_gi=i3,
              cornell48_gi_M=o.restir.ind_last["numSamples"].copy())
    np.savez_compressed(os.path.join(OUT, "frames_oracle.npz"), **fr)
    print("frames_oracle.npz", os.path.getsize(os.path.join(OUT, "frames_oracle.npz")), "bytes")


if __name__ == "__main__":
    main()
