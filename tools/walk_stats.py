#!/usr/bin/env python3
"""tools/walk_stats.py -- wave-level counters of the shadow-ray walk on the bench workload.
Needs a measurement build:  make -C restir_amd/csrc EXTRA=-DRS_WALK_STATS OUT=../librestir_stats.so VIEWER=
and RESTIR_HIP_LIB=restir_amd/librestir_stats.so."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from restir_amd import capi, scenes
from restir_amd.tiling import HipBackend, StripRenderer

W, H = 1920, 1080
capi.init(0)
sd = scenes.bistro_class(seed=2, scale=1.0) if (len(sys.argv) > 1 and sys.argv[1] == "bistro") else scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
backend = HipBackend(capi, scene, cam, W, H)
strips = StripRenderer(backend, 1, 0, H)
for _ in range(6):
    strips.frame(3, 0)
out = (C.c_ulonglong * 64)()
L = capi.lib()
L.rs_debug_walk_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert L.rs_debug_walk_stats(scene.handle, out, 1) == 0
if hasattr(L, "rs_debug_walk_stats_ordered"):
    L.rs_debug_walk_stats_ordered.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert L.rs_debug_walk_stats_ordered(scene.handle, (C.c_ulonglong * 32)(), 1) == 0
frames = 4
for _ in range(frames):
    strips.frame(3, 0)
assert L.rs_debug_walk_stats(scene.handle, out, 1) == 0
names = ["waves", "iterations", "leaf rounds", "triangle iterations", "verify iterations", "walk iterations",
         "walking lanes (sum)", "verifying lanes (sum)", "iterations without a walker", "triangle lane-tests"]
w = out[0]
print("per wave, mean over %d frames (%d waves/frame):" % (frames, w // frames))
for i, n in enumerate(names):
    print("  %-28s %10.2f" % (n, out[i] / w))
print("  walking lanes per walk iteration %.1f, verifying lanes per verify iteration %.1f, lanes per triangle iteration %.1f" %
      (out[6] / max(out[5], 1), out[7] / max(out[4], 1), out[9] / max(out[3], 1)))
if out[10] + out[12]:
    print("per RAY of the shadow-tree walk: %.3f occluded; node steps %.1f for an occluded ray, %.1f for an unoccluded one; triangle tests %.2f per ray" %
          (out[10] / (out[10] + out[12]), out[11] / max(out[10], 1), out[13] / max(out[12], 1), out[14] / (out[10] + out[12])))
if os.environ.get("WALK_STATS_TIME"):          # a -DRS_WALK_STATS_TIME build: slots 44.. hold the walk by iteration index instead of by depth
    print("shadow-tree walk by iteration index (buckets of 24 iterations): waves still iterating (per wave), walking lanes per iteration")
    for b in range(10):
        if out[54 + b]:
            print("   iterations %3d-%-4s waves %.3f  lanes %.1f" % (b * 24, str(b * 24 + 23) if b < 9 else "...", out[54 + b] / 24 / w if b < 9 else out[54 + b] / w, out[44 + b] / out[54 + b]))
    tot = 0
else:
    tot = sum(out[44:64])
if tot:
    print("shadow-tree steps by node depth (19 = 19 and deeper), fraction and cumulative:")
    acc = 0
    for d in range(20):
        acc += out[44 + d]
        print("   depth %2d  %6.3f  %6.3f" % (d, out[44 + d] / tot, acc / tot))
pw = out[16]
print("closest-hit packet walks (G-buffer + primary), per wave: union nodes mean %.1f max %d, orders %.2f, waves on the special-case path %.4f" %
      (out[17] / pw, out[18], out[19] / pw, out[20] / pw))
print("  of the union nodes: some lane passes the distance part %.1f, some lane enters %.1f, of which leaves %.1f (fast form only)" % (out[21] / pw, out[22] / pw, out[23] / pw))
print("  nodes at which every lane that passed the distance part clears the overlap part by 2^-19 * tRoot: %.1f" % (out[15] / pw))
print("  histogram of union nodes per wave (log2 buckets):", " ".join("%d:%d" % (1 << b, out[24 + b]) for b in range(20) if out[24 + b]))

try:
    o2 = (C.c_ulonglong * 32)()
    assert L.rs_debug_walk_stats_ordered(scene.handle, o2, 0) == 0
    if o2[23]:
        print("  the slowest ray of a wave visits %.1f nodes on average (fast form only); waves with >= 1024 union nodes: %d, union %.0f, slowest ray %.0f nodes" %
              (o2[23] / pw, o2[20], o2[21] / max(o2[20], 1), o2[22] / max(o2[20], 1)))
except AttributeError:
    pass
