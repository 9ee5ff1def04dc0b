#!/usr/bin/env python3
"""tools/asan/fuzz_scene_loader.py [seed] [pictures] -- the scene-file front end (scene text, OBJ reader, PNG / JPEG / TGA / HDR / PPM
decoders: host C++ of librestir_hip, restir_amd/csrc/scene_file.cpp) under AddressSanitizer + UBSan on the CPU: builds a
harness with g++ (no GPU, no hipcc) and feeds it the committed fixture files after random truncation, byte flips, insertions
and deletions.  Every file must either load or be refused with an error; any sanitizer report fails the run."""
import os, sys, subprocess, tempfile, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HARNESS = os.path.join(tempfile.gettempdir(), 'rs_scene_loader_asan')
subprocess.check_call(['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-omit-frame-pointer', '-D__HIP_PLATFORM_AMD__',
                       '-I/opt/rocm/include', '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'restir_amd', 'csrc'), '-x', 'c++',
                       os.path.join(ROOT, 'tools', 'asan', 'scene_loader_harness.cpp'), os.path.join(ROOT, 'restir_amd', 'csrc', 'scene_file.cpp'),
                       os.path.join(ROOT, 'restir_amd', 'csrc', 'scene_build.cpp'), '-o', HARNESS])
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'scene_files.npz'))
d = tempfile.mkdtemp()
from tests import scene_file_cases as cases
for name in cases.CASE_FILES:
    open(os.path.join(d, name), 'wb').write(g['file_' + name].tobytes())
pics = {}
for k in g.files:
    if k.startswith('jpeg_file_'): pics[k[10:] + '.jpg'] = g[k].tobytes()
    if k.startswith('tga_file_'): pics[k[9:] + '.tga'] = g[k].tobytes()
    if k.startswith('bmp_file_'): pics[k[9:] + '.bmp'] = g[k].tobytes()
for name in cases.CASE_FILES:
    if name.endswith(('.png', '.hdr', '.ppm')): pics[name] = g['file_' + name].tobytes()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cam = "Camera\nResolution 8 8\nFovY 20\nLensRadius 0\nFocalDist 1\nApertureMask Null\nSample 1\nDepth 1\nFile x\nEye 0 0 3\nRotation -90 0 0\nUp 0 1 0\n\n"
scenes = []
def mutate(b):
    b = bytearray(b)
    kind = rng.integers(0, 4)
    if kind == 0 and len(b) > 4: b = b[:rng.integers(1, len(b))]
    elif kind == 1:
        for _ in range(int(rng.integers(1, 8))): b[rng.integers(0, len(b))] = rng.integers(0, 256)
    elif kind == 2:
        i = rng.integers(0, len(b)); b[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 16)), dtype=np.uint8))
    else:
        i = rng.integers(0, len(b)); j = min(len(b), i + int(rng.integers(1, 32))); del b[i:j]
    return bytes(b)
n = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 400):
    name = list(pics)[it % len(pics)]
    ext = os.path.splitext(name)[1]
    p = os.path.join(d, f"m{it}{ext}")
    open(p, 'wb').write(mutate(pics[name]))
    s = os.path.join(d, f"s{it}.txt")
    open(s, 'w').write(f"Material m\nType Lambertian\nBaseColor {p}\nMetallic 0\nRoughness 1\nIor 1.5\nNormalMap Null\n\nObject o\n{d}/cube.obj\nMaterial m\nScale 1 1 1\n\n" + cam)
    scenes.append(s)
# mutated OBJ and scene texts
for it in range(200):
    src = ['cube.obj', 'poly.obj', 'numbers.obj', 'scene.txt'][it % 4]
    p = os.path.join(d, f"x{it}" + os.path.splitext(src)[1])
    open(p, 'wb').write(mutate(g['file_' + src].tobytes()))
    if src.endswith('.obj'):
        s = os.path.join(d, f"t{it}.txt"); open(s, 'w').write(f"Object o\n{p}\nMaterial Null\nScale 1 1 1\n\n" + cam); scenes.append(s)
    else:
        shutil.copy(p, os.path.join(d, f"scene_m{it}.txt")); scenes.append(os.path.join(d, f"scene_m{it}.txt"))
for i in range(0, len(scenes), 50):
    r = subprocess.run([HARNESS] + scenes[i:i + 50], capture_output=True, text=True, cwd=d, timeout=300)
    if r.returncode != 0 or 'ERROR' in r.stderr or 'runtime error' in r.stderr:
        print("FAIL batch", i, r.returncode); print(r.stderr[-3000:]); sys.exit(1)
    print(r.stdout.strip())
print("no sanitizer report")
