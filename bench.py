#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: Mrays/s and ms/frame of the ReSTIR-DI per-frame
sequence (runCuda, src/main.cpp:146-185: GBuffer::render -> ReSTIRDirect -> copyImageToPBO ->
GBuffer::update) at 1920x1080, 32 candidates, spatiotemporal reuse, on the procedural Sponza-class
scene (262 144 triangles, 1 024 emissive; BASELINE config 3).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A step = one frame.  A ray = one BVH walk (intersect or testOcclusion call): 1 G-buffer ray + 1
shading ray per pixel + 1 shadow ray per shaded pixel; counted by the kernels.

N > 1: the 1920x1080 framebuffer is cut into N row strips (strong scaling, fixed total work).  Every
rank renders its strip (+5 G-buffer halo rows), exchanges 5 rows of published reservoirs with its
strip neighbours over RCCL point-to-point between phase A and phase B, tone-maps its strip and the
RGBA8 strips are gathered on rank 0 -- all inside the timed region.

One JSON line is printed by rank 0; besides the contract's fields it carries
  roofline      the spatial-reuse pass (k_spatial_shade): algorithmic 92 B/px over its
                HIP-event duration, against the 8 TB/s HBM3E peak
  cpu_baseline  the CPU oracle (oracle/restir_oracle.c, OpenMP) on this box's host cores, on a
                bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 1920, 1080
REUSE = 3                      # ReservoirReuse::Spatiotemporal
TONEMAP = 2                    # ToneMapping::ACES (Settings default, src/common.cpp:4)
HALO = 5
ALGO_BYTES_PER_PIXEL = 92      # spatial-reuse pass
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def strip_bounds(height, n, rank):
    """Contiguous row strips; strip heights differ by at most one row."""
    base, rem = divmod(height, n)
    y0 = rank * base + min(rank, rem)
    return y0, y0 + base + (1 if rank < rem else 0)


def cpu_baseline(sd, frames):
    """The oracle on the host cores: `frames` full 1920x1080 spatiotemporal frames (bounded sample)."""
    from tests.common import OracleRenderer
    threads = len(os.sched_getaffinity(0))
    o = OracleRenderer(sd, WIDTH, HEIGHT)
    o.frame(REUSE)                                   # untimed: thread-pool start-up, page faults
    rays = 0
    t0 = time.perf_counter()
    for _ in range(frames):
        o.frame(REUSE)
        rays += o.rays + WIDTH * HEIGHT
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "ms_per_frame": dt / frames * 1e3,
            "sample": f"{frames} frames of the same workload (1920x1080 spatiotemporal, Sponza-class 262144 tris), OpenMP over rows"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--cpu-frames", type=int, default=3, help="frames timed for cpu_baseline (0 = skip)")
    ap.add_argument("--orbit", action="store_true", help="orbit the camera (runCuda animateCamera) instead of the static default")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; the HIP path has no fallback"
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from restir_amd import capi, scenes
    from restir_amd.scenes import orbit_position
    capi.init(local_rank)

    sd = scenes.sponza_class(seed=1, scale=1.0)
    scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
    cam = capi.camera_update(sd.camera(WIDTH, HEIGHT))
    gbuf = capi.GBuffer(WIDTH, HEIGHT)
    restir = capi.ReSTIR(WIDTH, HEIGHT)
    restir.enable_timing(True)
    image = torch.zeros((WIDTH * HEIGHT, 3), dtype=torch.float32, device="cuda")
    pbo = torch.zeros((WIDTH * HEIGHT, 4), dtype=torch.uint8, device="cuda")

    y0, y1 = strip_bounds(HEIGHT, world, rank)
    gy0, gy1 = max(0, y0 - HALO), min(HEIGHT, y1 + HALO)
    up, down = rank - 1, rank + 1
    if world > 1:
        nb = restir.halo_bytes(HALO)
        send_up = torch.empty(nb, dtype=torch.uint8, device="cuda"); recv_up = torch.empty_like(send_up)
        send_dn = torch.empty(nb, dtype=torch.uint8, device="cuda"); recv_dn = torch.empty_like(send_dn)
        # ragged strips: gather buffers sized to the largest strip
        max_rows = max(strip_bounds(HEIGHT, world, r)[1] - strip_bounds(HEIGHT, world, r)[0] for r in range(world))
        gather_out = [torch.empty((max_rows * WIDTH, 4), dtype=torch.uint8, device="cuda") for _ in range(world)] if rank == 0 else None
        strip_rgba = torch.zeros((max_rows * WIDTH, 4), dtype=torch.uint8, device="cuda")

    base_pos = sd.camera_args["position"]
    state = {"looper": 0}
    capi.set_sync(False)                   # launches are enqueued; the timed region is bracketed by synchronize()

    def frame():
        if args.orbit:
            p = orbit_position(base_pos, state["looper"], radius=1.0)
            for i in range(3):
                cam.position[i] = float(p[i])
            capi.camera_update(cam)
        if world == 1:
            gbuf.render(scene, cam)
            restir.direct(scene, cam, gbuf, image.data_ptr(), 0, state["looper"], REUSE)
            capi.copy_image_to_pbo(pbo.data_ptr(), image.data_ptr(), WIDTH, HEIGHT, TONEMAP, 1.0)
        else:
            gbuf.render(scene, cam, gy0, gy1)
            restir.phase_a(scene, cam, gbuf, state["looper"], REUSE, y0, y1)
            ops = []
            if up >= 0:
                restir.halo_pack(y0, HALO, send_up.data_ptr())
                ops += [dist.P2POp(dist.isend, send_up, up), dist.P2POp(dist.irecv, recv_up, up)]
            if down < world:
                restir.halo_pack(y1 - HALO, HALO, send_dn.data_ptr())
                ops += [dist.P2POp(dist.isend, send_dn, down), dist.P2POp(dist.irecv, recv_dn, down)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            if up >= 0:
                restir.halo_unpack(y0 - HALO, HALO, recv_up.data_ptr())
            if down < world:
                restir.halo_unpack(y1, HALO, recv_dn.data_ptr())
            restir.phase_b(scene, cam, gbuf, image.data_ptr(), 0, REUSE, y0, y1)
            restir.end_frame()
            rows = y1 - y0
            capi.copy_image_to_pbo(strip_rgba.data_ptr(), image.data_ptr() + y0 * WIDTH * 12, WIDTH, rows, TONEMAP, 1.0)
            dist.gather(strip_rgba, gather_out, dst=0)
        state["looper"] += 1
        gbuf.update(cam)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        frame()
    barrier()
    spatial_ms = []
    pass_ms = np.zeros(4)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    barrier()
    elapsed = time.perf_counter() - t0

    # per-pass times of a few extra (untimed) frames, measured with HIP events on the library's stream
    for _ in range(min(20, max(3, args.steps))):
        frame()
        torch.cuda.synchronize()
        ms = restir.pass_times()
        spatial_ms.append(ms[3]); pass_ms += np.array(ms)
    pass_ms /= len(spatial_ms)
    torch.cuda.synchronize()

    rows = y1 - y0
    local_rays = restir.ray_total(min(args.steps, 1024)) / min(args.steps, 1024) * args.steps + (gy1 - gy0) * WIDTH * args.steps
    t = torch.tensor([elapsed, float(local_rays)], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, total_rays = float(tmax[0]), float(tsum[1])
    else:
        total_rays = float(local_rays)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        spatial_us = float(np.median(spatial_ms)) * 1e3
        algo_bytes = ALGO_BYTES_PER_PIXEL * WIDTH * rows
        achieved = algo_bytes / (spatial_us * 1e-6) / 1e9
        out = {
            "metric": "Mrays/s (1920x1080 ReSTIR-DI, 32 candidates, spatiotemporal reuse)",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 3: procedural Sponza-class seed 1 (262144 triangles, 1024 emissive), 1920x1080, "
                                   "32 RIS candidates, spatiotemporal ReSTIR-DI, frame = GBuffer::render + ReSTIRDirect + copyImageToPBO",
                       "camera": "orbit" if args.orbit else "static", "tiling": f"{world} row strips, 5-row reservoir halo over RCCL p2p" if world > 1 else "none",
                       "rays_per_frame": total_rays / args.steps},
            "roofline": {"bound": "hbm", "kernel": "k_spatial_shade", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes": algo_bytes, "kernel_us": spatial_us},
            "pass_ms": {"primary": float(pass_ms[0]), "ris": float(pass_ms[1]), "shadow_temporal": float(pass_ms[2]), "spatial_shade": float(pass_ms[3])},
        }
        if world == 1 and args.cpu_frames > 0:
            out["cpu_baseline"] = cpu_baseline(sd, args.cpu_frames)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
