#!/usr/bin/env python3
"""tools/ris_stats.py [bistro] -- of the 32 RIS candidates per shaded pixel (restir.cu:155-170), how many leave
sampleDirectLightNoVisibility without a valid pdf (a single-sided light facing away, scene.h:409) and how many more end with a zero
weight (the light below the surface's horizon): lanes that still execute the pdf conversion, the BSDF and the weight's divisions
because their wave does.  Needs a measurement build:  tools/build_variant.sh stats -DRS_WALK_STATS
and RESTIR_HIP_LIB=restir_amd/librestir_stats.so."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from restir_amd import capi, scenes
from restir_amd.tiling import HipBackend, StripRenderer

W, H = 1920, 1080
capi.init(0)
bistro = len(sys.argv) > 1 and sys.argv[1] == "bistro"
sd = scenes.bistro_class(seed=2, scale=1.0) if bistro else scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
strips = StripRenderer(HipBackend(capi, scene, cam, W, H), 1, 0, H)
L = capi.lib()
L.rs_debug_walk_stats_ordered.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
out = (C.c_ulonglong * 32)()
assert L.rs_debug_walk_stats_ordered(scene.handle, out, 1) == 0
frames = 4
for _ in range(frames):
    strips.frame(3, 0)
assert L.rs_debug_walk_stats_ordered(scene.handle, out, 1) == 0
cand, inv, zero, steps, all_inv, all_dead = (out[24 + i] for i in range(6))
print("%s-class scene, 1080p, %d frames: %.1f M candidates per frame" % ("Bistro" if bistro else "Sponza", frames, cand / frames / 1e6))
print("  without a valid pdf (light faces away)   %.3f" % (inv / cand))
print("  valid pdf, weight 0 (below the horizon)   %.3f" % (zero / cand))
print("  wave-level candidate steps %d per frame; all lanes without a pdf in %.4f of them, all lanes dead (no pdf or zero weight) in %.4f" %
      (steps // frames, all_inv / steps, all_dead / steps))
