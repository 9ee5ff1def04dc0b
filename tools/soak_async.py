#!/usr/bin/env python3
"""tools/soak_async.py [minutes] -- randomized check that asynchronous launches with overlapped frames (rs_set_sync(0): G-buffer
render and primary-ray + RIS kernels on the auxiliary streams, ring / double buffers, event ordering) give the images of the
synchronous mode bit for bit: random scene, frame size, reuse mode, camera path, strips / bands, extra renders, EAW filter,
tone map, path-tracing passes in between; since round 6 also the filter and the tone map of its result on the denoise stream
(rs_set_denoise_stream: with the caller's own copy behind rs_join_denoise_stream, or with no join at all so that only the display
images tell) and host waits after random frames (a frame that finds the device idle takes the synchronous mode's launch forms).
No oracle involved (the synchronous mode is what the parity tests pin)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from restir_amd import capi, scenes
from restir_amd.scenes import orbit_position
from restir_amd.tiling import HipBackend

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
capi.init(0)
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "7")))
pool = {}
def scene_of(name):
    if name not in pool:
        sd = {"cornell": scenes.cornell_box, "textured": lambda: scenes.cornell_textured(env=True),
              "sponza": lambda: scenes.sponza_class(1, 0.08), "sponza_big": lambda: scenes.sponza_class(1, 0.5)}[name]()
        pool[name] = (sd, capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials, textures=sd.textures, env_map_tex=sd.env_map_tex))
    return pool[name]

def run(cfg, overlapped):
    sd, scene = scene_of(cfg["scene"])
    W, H = cfg["size"]
    cam = capi.camera_update(sd.camera(W, H))
    b = HipBackend(capi, scene, cam, W, H)
    eaw = capi.EAWFilter(W, H, 5) if cfg["eaw"] else None
    out = torch.zeros_like(b.image)
    imgs = []
    capi.set_sync(not overlapped)
    capi.set_side_stream(cfg.get("streams", 1))
    ds = cfg.get("denoise_stream", 0) if (overlapped and eaw is not None) else 0
    capi.set_denoise_stream(1 if ds else 0)
    try:
        for f in range(cfg["frames"]):
            if cfg["orbit"]:
                p = orbit_position(sd.camera_args["position"], f, radius=0.4)
                for i in range(3): cam.position[i] = float(p[i])
                capi.camera_update(cam)
            b.gbuffer_render(0, H)
            if f in cfg["extra_render"]: b.gbuffer_render(0, max(8, H // 3))
            if cfg["bands"]:
                cut = cfg["bands"]
                b.phase_a(f, cfg["reuse"], 0, cut); b.phase_a(f, cfg["reuse"], cut, H)
                b.phase_b(f % 3, cfg["reuse"], cut, H); b.phase_b(f % 3, cfg["reuse"], 0, cut)
            else:
                b.phase_a(f, cfg["reuse"], 0, H); b.phase_b(f % 3, cfg["reuse"], 0, H)
            imgs.append(b.image.clone())
            shown = b.image.data_ptr()
            if eaw is not None:
                p = eaw.filter(out.data_ptr(), b.image.data_ptr(), b.gbuf, cam)
                if cfg.get("denoise_stream", 0) != 2:               # (2: no copy of the caller's own, no join: the display image below is the witness)
                    if ds: capi.join_denoise_stream()
                    t = torch.empty_like(b.image); capi.hip_memcpy_d2d_async(t.data_ptr(), p, t.numel() * 4); imgs.append(t)
                if cfg.get("denoise_stream", 0): shown = p          # the tone map reads the filter's result (on the denoise stream when there is one)
            if f in cfg["pt"]:
                t = torch.zeros_like(b.image); capi.path_trace_direct(scene, cam, t.data_ptr(), 0, f); imgs.append(t)
            pbo = torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda")
            capi.copy_image_to_pbo(pbo.data_ptr(), shown, W, H, 2, 1.0); imgs.append(pbo)
            b.end_frame()
            if f in cfg.get("host_waits", ()): capi.synchronize()
        capi.synchronize(); torch.cuda.synchronize()
    finally:
        capi.set_sync(True)
        capi.set_side_stream(4)
        capi.set_denoise_stream(0)
    res = [t.cpu().numpy() for t in imgs] + [b.restir.download(1).view(np.uint8)]
    if eaw is not None: eaw.destroy()
    return res

t_end = time.time() + minutes * 60
n = bad = 0
last = time.time()
while time.time() < t_end:
    H = int(rng.integers(16, 400)); W = int(rng.integers(16, 640))
    frames = int(rng.integers(2, 9))
    cfg = dict(streams=int(rng.choice([1, 3, 4])), scene=str(rng.choice(["cornell", "textured", "sponza", "sponza_big"])), size=(W, H), frames=frames, reuse=int(rng.integers(0, 4)),
               orbit=bool(rng.integers(0, 2)), eaw=bool(rng.integers(0, 3) == 0), bands=int(rng.integers(8, H - 1)) if H > 24 and rng.integers(0, 3) == 0 else 0,
               extra_render=set(rng.integers(0, frames, 1).tolist()) if rng.integers(0, 3) == 0 else set(), pt=set(rng.integers(0, frames, 1).tolist()) if rng.integers(0, 4) == 0 else set(),
               denoise_stream=int(rng.integers(0, 3)), host_waits=set(rng.integers(0, frames, int(rng.integers(0, 4))).tolist()) if rng.integers(0, 3) == 0 else set())
    a, o = run(cfg, False), run(cfg, True)
    same = len(a) == len(o) and all(x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a, o))
    n += 1
    if not same:
        bad += 1
        which = [(k, int((x.reshape(len(x), -1).view(np.uint8) != y.reshape(len(y), -1).view(np.uint8)).any(axis=1).sum()), len(x))
                 for k, (x, y) in enumerate(zip(a, o)) if x.shape == y.shape and not np.array_equal(x.view(np.uint8), y.view(np.uint8))]
        a2 = run(cfg, False)
        again = all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a, a2))
        print("MISMATCH", cfg, "outputs (index, differing rows, rows):", which, "| synchronous mode repeats itself:", again, flush=True)
    if time.time() - last > 30:
        print("%d configurations, %d mismatches" % (n, bad), flush=True); last = time.time()
print("done: %d configurations, %d mismatches" % (n, bad), flush=True)
sys.exit(1 if bad else 0)
