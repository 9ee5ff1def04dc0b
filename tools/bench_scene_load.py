#!/usr/bin/env python3
"""tools/bench_scene_load.py [scale] -- host-side throughput of the scene-file front end (rs_scene_file_load: scene text, OBJ
reader, instance baking) on the Sponza-class benchmark scene exported with restir_amd.scene_io, next to the reference's own
OBJ loader (tinyobj::LoadObj + the flattening loop of Resource::loadOBJMesh through oracle/_ref/libref_loaders.so, build
container only) on the same files.  No GPU needed."""
import ctypes as C
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from restir_amd import capi, scene_io, scenes

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sd = scenes.sponza_class(seed=1, scale=scale)
with tempfile.TemporaryDirectory() as d:
    t0 = time.perf_counter()
    path = scene_io.export_scene_data(sd, d, 1920, 1080, name="sponza")
    objs = sorted(f for f in os.listdir(d) if f.endswith(".obj"))
    nbytes = sum(os.path.getsize(os.path.join(d, f)) for f in objs)
    print("exported %d triangles as %d OBJ files, %.1f MB of text (%.1f s)" % (sd.num_prims, len(objs), nbytes / 1e6, time.perf_counter() - t0))
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        a = capi.SceneFile(path)
        best = min(best, time.perf_counter() - t0)
    assert a.vertices.shape[0] == sd.num_prims and np.array_equal(a.vertices, sd.vertices)
    print("rs_scene_file_load (text + OBJ + baking, incl. the copy into numpy): %.3f s = %.0f MB/s, %.2f M triangles/s" %
          (best, nbytes / best / 1e6, sd.num_prims / best / 1e6))
    try:
        from oracle import binding as ob
        R = ob.ref_loaders()
    except Exception:
        R = None
    if R is not None:
        best = 1e9
        cap = sd.num_prims * 3
        v = np.zeros((cap, 3), np.float32); n = np.zeros((cap, 3), np.float32); t = np.zeros((cap, 2), np.float32)
        for _ in range(3):
            t0 = time.perf_counter()
            total = 0
            for f in objs:
                total += R.ref_obj_load(os.path.join(d, f).encode(), cap, v.ctypes.data, n.ctypes.data, t.ctypes.data)
            best = min(best, time.perf_counter() - t0)
        assert total == sd.num_prims * 3
        print("reference loader (tinyobj::LoadObj + flatten, OBJ files only):        %.3f s = %.0f MB/s, %.2f M triangles/s" %
              (best, nbytes / best / 1e6, sd.num_prims / best / 1e6))
    else:
        print("reference loader: oracle/_ref/libref_loaders.so not built here")
