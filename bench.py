#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: Mrays/s and ms/frame of the ReSTIR-DI per-frame
sequence (runCuda, src/main.cpp:146-185: GBuffer::render -> ReSTIRDirect -> copyImageToPBO ->
GBuffer::update) at 1920x1080, 32 candidates, spatiotemporal reuse, on the procedural Sponza-class
scene (262 144 triangles, 1 024 emissive; BASELINE config 3).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A step = one frame.  A ray = one BVH walk (intersect or testOcclusion call): 1 G-buffer ray + 1
shading ray per pixel + 1 shadow ray per shaded pixel; counted by the kernels.

N > 1: the 1920x1080 framebuffer is cut into N row strips (strong scaling: the total work is fixed).
Every rank renders its strip, exchanges 5 border rows of published reservoirs and of the G-buffer id / normal / depth planes
with its strip neighbours over RCCL point-to-point between phase A and phase B, tone-maps its strip, and the RGBA8 strips are
gathered on rank 0 (asynchronously: the gather of one frame overlaps the next frame's kernels; the last gathers are waited for
before the clock stops) -- all inside the timed region.  The frames go through the PRODUCT's strip driver, the C ABI of
restir_amd/csrc/strips.hip (rs_strips_frame / rs_strips_gather_begin / _end over rs_comm_create_rccl_lib), on an ncclComm_t this
script creates the way a C++ caller does (ncclGetUniqueId on rank 0, the id broadcast, ncclCommInitRank; restir_amd/rccl.py);
torch.distributed (gloo) is the control plane only: the id broadcast, barriers, the reduction of the timings.
  BENCH_STRIP_DRIVER=py      the Python form of the same schedule (restir_amd/tiling.py over torch.distributed) for cross-checks
  BENCH_STRIP_TRANSPORT=gloo the C driver over host callbacks + gloo: rehearsal on a one-GPU box (with BENCH_DEVICE=0 every rank
                             uses the same card; RCCL refuses two ranks on one device)
  BENCH_FORCE_STRIPS=1       N = 1 through the strip driver and a one-rank ncclComm as well

One JSON line is printed by rank 0; besides the contract's fields it carries
  roofline      the spatial-reuse pass (k_spatial_shade): algorithmic 92 B/px over its
                HIP-event duration (events recorded on the stream the kernel is launched on),
                against the 8 TB/s HBM3E peak; `traffic` is filled from profiles/ PMC runs when known
  cpu_reference_loop  closest-hit Mrays/s of a host loop over the reference's own compiled intersection code (primary rays only)
  cpu_baseline  the CPU oracle (oracle/restir_oracle.c, OpenMP) on this box's host cores, on a
                bounded sample of the same workload (rank 0, N = 1 only)
  parity        N = 1: the frames cpu_baseline rendered, rendered again by the GPU and compared bit for bit (differing_pixels, mean_l1);
                N > 1: six frames through the strip driver from fresh reservoirs, gathered on rank 0 and compared bit for bit with rank 0's
                own full-frame render of the same frames (`strips_parity` when BENCH_FORCE_STRIPS=1 sends N = 1 through the strip driver)
  cpu_config1   BASELINE config 1 (Cornell 256x256, 1 spp PTDirect) as a host loop on the same cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 1920, 1080
REUSE = 3                      # ReservoirReuse::Spatiotemporal
TONEMAP = 2                    # ToneMapping::ACES (Settings default, src/common.cpp:4)
ALGO_BYTES_PER_PIXEL = 92      # spatial-reuse pass
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
# HBM bytes per launch from the PMC passes of tools/profile.sh on this same command (FETCH_SIZE x 2 + WRITE_SIZE,
# the gfx950 correction of MI355X_MICROARCH.md), condensed by tools/summarize_profile.py; counters cannot be read
# from inside the process, so `roofline.traffic` quotes the committed summary (null if it is absent).
PMC_SUMMARY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_final2_hbm_counters.json")


def pmc_traffic(kernel):
    """(bytes per launch, the kernel's average duration in that profile): the duration lets a reader see whether the committed
    summary still describes the kernel that ran here."""
    try:
        with open(PMC_SUMMARY) as fh:
            k = json.load(fh)["kernels"][kernel]
            return float(k["hbm_bytes_per_launch"]), (float(k["average_ns"]) / 1e3 if k.get("average_ns") else None)
    except (OSError, KeyError, ValueError):
        return None, None


OVERLAPPED_STATS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_final2_kernel_stats_overlapped.csv")


def overlapped_kernel_us(kernel):
    """Average duration of a kernel in the committed rocprofv3 kernel trace of this command with the frames overlapped."""
    try:
        import csv
        with open(OVERLAPPED_STATS) as fh:
            rows = [r for r in csv.reader(fh) if r and not r[0].startswith("#")]
        head = rows[0]
        for r in rows[1:]:
            if r[0] == kernel:
                return float(r[head.index("average_ns")]) / 1e3
    except (OSError, ValueError, IndexError):
        pass
    return None


def host_threads():
    """CPU share of this process: min(affinity, cgroup quota), capped at 64."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(sd, frames):
    """The oracle on the host cores: `frames` full 1920x1080 spatiotemporal frames (bounded sample).  Also returns the images of
    all frames it rendered (the untimed first one included): the checker's side of the `parity` field."""
    threads = host_threads()
    os.environ["OMP_NUM_THREADS"] = str(threads)          # read by libgomp when liboracle.so is loaded
    from oracle import binding as ob
    from tests.common import OracleRenderer
    ob.set_libm_mode(1)                                   # cos / sin of the spatial taps correctly rounded: the mode the product is exact in
    o = OracleRenderer(sd, WIDTH, HEIGHT)
    images = [o.frame(REUSE).copy()]                      # untimed: thread-pool start-up, page faults
    rays = 0
    dt = 0.0
    for _ in range(frames):
        t0 = time.perf_counter()
        o.frame(REUSE)
        dt += time.perf_counter() - t0
        rays += o.rays + WIDTH * HEIGHT
        images.append(o.image.copy())
    ob.set_libm_mode(0)
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "ms_per_frame": dt / frames * 1e3,
            "sample": f"{frames} frames of the same workload (1920x1080 spatiotemporal, Sponza-class 262144 tris), OpenMP over rows"}, images


def parity_against(capi, sd, scene, oracle_images):
    """The frames cpu_baseline rendered (looper 0, 1, ...: static camera, spatiotemporal reuse) once more on the GPU, from fresh
    reservoirs, in the reference's synchronous mode, compared pixel by pixel: the checker's verdict on the workload that was timed."""
    import numpy as np
    from tests.common import HipRenderer
    capi.set_sync(True)
    h = HipRenderer(capi, sd, WIDTH, HEIGHT, scene=scene)
    differing, l1 = 0, 0.0
    for ref in oracle_images:
        got = h.frame(REUSE)
        differing += int(np.count_nonzero((ref.view(np.uint32) != got.view(np.uint32)).any(axis=1)))
        l1 += float(np.abs(ref.astype(np.float64) - got.astype(np.float64)).sum(axis=1).mean())
    capi.set_sync(False)
    return {"frames": len(oracle_images), "pixels_per_frame": WIDTH * HEIGHT, "differing_pixels": differing, "mean_l1": l1 / len(oracle_images),
            "checker": "oracle/restir_oracle.c (libm mode: correctly rounded), frames looper 0.. of the benchmark workload from fresh reservoirs, "
                       "radiance compared bit for bit"}


def cpu_config1(threads, seconds=3.0):
    """BASELINE config 1: Cornell box, 256x256, 1 spp raw path trace without ReSTIR (PTDirect semantics, looper 0, 1, ...)
    as a host loop -- the oracle's restatement of PTDirectKernel (src/pathtrace.cu:279-328) over the reference's intersection
    code, OpenMP over rows; no GPU involved."""
    import numpy as np
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle import binding as ob
    from restir_amd import scenes
    from tests.common import oracle_scene
    sd = scenes.cornell_box()
    w = h = 256
    scene = oracle_scene(sd)
    cam = ob.camera_update(sd.camera(w, h))
    img = np.zeros((w * h, 3), np.float32)
    ob.pt_direct(scene, cam, img, 0, 0)                    # untimed: thread-pool start-up
    rays, frames = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        rays += ob.pt_direct(scene, cam, img, 0, frames)
        frames += 1
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "ms_per_frame": dt / frames * 1e3,
            "sample": f"{frames} frames of Cornell box 256x256, 1 spp pathTraceDirect (primary + one shadow ray per shaded pixel), host loop only"}


def cpu_reference_loop(sd, threads):
    """The baseline north_star names: a host-side loop over the reference's own intersections.h / bvh.h -- DevScene::intersect's
    traversal around the reference's compiled AABB::intersect / intersectTriangle on the reference builder's tree
    (oracle/_ref/libref_subset.so, prebuilt in the build container; oracle/ref_subset.cpp).  Closest hits of the benchmark view's
    camera rays (one jittered ray per pixel of the 1920x1080 frame), repeated for about five seconds.  None when the library
    is not there."""
    import ctypes as C
    import numpy as np
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle import binding as ob
    R = ob.ref_subset()
    if R is None:
        return None
    w, h = WIDTH, HEIGHT
    cam = ob.camera_update(sd.camera(w, h))
    rng = np.random.default_rng(1)
    ys, xs = np.mgrid[0:h, 0:w]
    xy = np.stack([xs.reshape(-1), ys.reshape(-1)], 1).astype(np.int32)
    r4 = rng.uniform(0, 1, (len(xy), 4)).astype(np.float32)
    rays = np.zeros((len(xy), 6), np.float32)
    ob.lib().orc_camera_sample(C.byref(cam), len(xy), xy.reshape(-1), r4.reshape(-1), rays.reshape(-1))
    # the reference's BVHBuilder prints its progress on stdout (src/bvh.cpp:15): send it to stderr, stdout carries one JSON line
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        prim, _, seconds = ob.ref_closest_hit_loop(R, sd.vertices, rays, min_seconds=5.0)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    return {"value": len(xy) / seconds / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "reference",
            "hit_fraction": float((prim >= 0).mean()),
            "sample": f"closest hit of the {len(xy)} camera rays of the benchmark view, repeated for 5 s, through the reference's compiled "
                      "AABB::intersect / intersectTriangle on the reference builder's MTBVH, OpenMP over rays"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cpu-frames", type=int, default=3, help="frames timed for cpu_baseline (0 = skip)")
    ap.add_argument("--orbit", action="store_true", help="orbit the camera (runCuda animateCamera) instead of the static default")
    args = ap.parse_args()
    # A rank that waits for a peer that will never answer would hang the whole job: with N > 1 every rank ends itself (Python
    # stacks of all threads on stderr) if the run takes longer than BENCH_WATCHDOG seconds (default 900; N = 1: only when set).
    watchdog = os.environ.get("BENCH_WATCHDOG", "900" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "")
    if watchdog:
        import faulthandler
        faulthandler.dump_traceback_later(float(watchdog), exit=True)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run, one rank per GPU")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; the HIP path has no fallback"
    # which strip driver runs the frames: "c" = rs_strips_* of librestir_hip (the product; default for N > 1), "py" = restir_amd/tiling.py,
    # "none" = the plain single-GPU calls (default for N = 1)
    force_strips = os.environ.get("BENCH_FORCE_STRIPS", "0") == "1"
    driver = os.environ.get("BENCH_STRIP_DRIVER", "c" if (world > 1 or force_strips) else "none")
    if world > 1 and driver == "none":
        driver = "c"
    transport = os.environ.get("BENCH_STRIP_TRANSPORT", "rccl")
    # rehearsal of the N > 1 path on a one-GPU box: BENCH_DEVICE=0 puts every rank on one card
    device = int(os.environ.get("BENCH_DEVICE", local_rank))
    # control plane: gloo for the C driver (its data path is the library's own RCCL transport); the Python driver's data path IS
    # torch.distributed, so it takes nccl (= RCCL) unless told otherwise
    backend_name = os.environ.get("BENCH_DIST_BACKEND", "gloo" if driver == "c" else "nccl")
    torch.cuda.set_device(device)
    if world > 1:
        if backend_name == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend_name)
    ctl_device = "cuda" if (world > 1 and backend_name == "nccl") else "cpu"      # where control-plane tensors live

    from restir_amd import capi, scenes
    from restir_amd.scenes import orbit_position
    from restir_amd.tiling import HipBackend, StripRenderer, calibrate_bounds, strip_bounds
    capi.init(device)

    sd = scenes.sponza_class(seed=1, scale=1.0)
    scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
    # BENCH_SOBOL=1: the reference built with SAMPLER_USE_SOBOL true (src/sampler.h:9-36) -- the 10 000 x 200 table of
    # restir_amd/sobol.py on the scene, looper wrapped at 10 000; the default (and the headline) is the default engine
    sobol_num = None
    if os.environ.get("BENCH_SOBOL", "0") == "1":
        from restir_amd import sobol
        table = sobol.sobol_table()
        scene.set_sample_sequence(table)
        sobol_num = len(table)
    cam = capi.camera_update(sd.camera(WIDTH, HEIGHT))
    if driver == "c" and world > 1:
        # the strip driver enqueues its RCCL transfers on the library stream: give it a stream of the kind RCCL is always used with
        # (an ordinary non-blocking one, as torch's own process groups use) instead of the legacy default stream
        lib_stream = torch.cuda.Stream()
        torch.cuda.set_stream(lib_stream)
    backend = HipBackend(capi, scene, cam, WIDTH, HEIGHT)          # (hands torch's current stream to the library: rs_set_stream)
    capi.set_sync(False)                   # launches are only enqueued; the timed region is bracketed by synchronize()
    # N > 1: strip heights balanced by measured cost before the warm-up (rows near the horizon cost several times a sky row and
    # the slowest strip sets the frame time); BENCH_EVEN_STRIPS=1 keeps equal heights
    bounds = None
    if world > 1 and os.environ.get("BENCH_EVEN_STRIPS", "0") != "1":
        bounds = calibrate_bounds(backend, world, rank, HEIGHT, dist, torch.cuda.synchronize, reuse=REUSE)
        torch.cuda.synchronize()
        backend = HipBackend(capi, scene, cam, WIDTH, HEIGHT)          # fresh reservoirs and G-buffer for the measured run
    if bounds is None:
        bounds = [strip_bounds(HEIGHT, world, r) for r in range(world)]
    base_pos = sd.camera_args["position"]
    state = {"looper": 0, "frame_no": 0}

    def move_camera(looper):
        if args.orbit:
            p = orbit_position(base_pos, looper, radius=1.0)
            for i in range(3):
                cam.position[i] = float(p[i])
            capi.camera_update(cam)

    rccl = None
    fallback = None
    if driver == "c":
        # ---- the product's strip driver: strips.hip through the C ABI ---------------------------------------------------------
        from restir_amd.rccl import GlooTransport, RcclComm
        comm = None
        if transport == "rccl":
            def bcast(raw):
                if world == 1:
                    return raw
                box = [raw]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            err = ""
            try:
                rccl = RcclComm(rank, world, bcast)        # before anything else in this process touches RCCL
                comm = capi.Comm.rccl(rccl.handle.value, rank, world, rccl.path)
            except Exception as e:                         # no RCCL on this box, bootstrap refused ...: the ranks agree on what to do next
                err = repr(e)
            if world > 1:
                errs = [None] * world
                dist.all_gather_object(errs, err)
                err = next((e for e in errs if e), "")
            if err:
                # still the C driver, over host callbacks + the control plane: slow (staged through host memory), but it finishes and says so
                fallback = "RCCL communicator could not be created (%s): the strip driver runs over host callbacks + gloo" % err
                print("bench.py: " + fallback, file=sys.stderr, flush=True)
                transport, rccl, comm = "gloo", None, None
        if comm is None:
            comm = GlooTransport(capi, dist, torch).comm(rank, world)
        drv = capi.Strips(comm, WIDTH, HEIGHT, [b[0] for b in bounds] + [HEIGHT])
        y0, y1 = drv.y0, drv.y1
        assert (y0, y1) == tuple(bounds[rank])
        rows = y1 - y0
        # the display image, full-frame sized on every rank (rs_strips_gather addresses rows in place); two of them, so that the
        # gather of frame f travels while frame f + 1 renders
        pbos = [torch.zeros((HEIGHT * WIDTH, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]

        def frame(bk=None, st=None, pb=None):
            # (bk, st, pb: another set of render objects, frame counters and display buffers -- the parity check below runs this very
            # function on fresh ones)
            bk, st, pb = bk or backend, st or state, pb or pbos
            move_camera(st["looper"])
            drv.frame(bk.restir, scene, cam, bk.gbuf, bk.image.data_ptr(), 0, st["looper"], REUSE)
            bk.gbuf.update(cam)
            if args.orbit and world > 1:
                drv.exchange_history(bk.restir, bk.gbuf)
            st["looper"] = st["looper"] + 1 if sobol_num is None else (st["looper"] + 1) % sobol_num
            k = st["frame_no"] % 2; st["frame_no"] += 1
            drv.gather_end(k)                                           # the gather that read pb[k] two frames ago
            capi.copy_image_to_pbo(pb[k].data_ptr() + y0 * WIDTH * 4, bk.image.data_ptr() + y0 * WIDTH * 12, WIDTH, rows, TONEMAP, 1.0)
            drv.gather_begin(pb[k].data_ptr(), 4, 0, k)

        def finish_gathers():
            drv.gather_end(0); drv.gather_end(1)

        def pbo_ptr():
            return pbos[0].data_ptr() + y0 * WIDTH * 4
    else:
        # ---- restir_amd/tiling.py over torch.distributed (N = 1: the plain calls) -----------------------------------------------
        strips = StripRenderer(backend, world, rank, HEIGHT, dist=dist if world > 1 else None, share_history=args.orbit, bounds=bounds)
        y0, y1 = strips.y0, strips.y1
        rows = y1 - y0
        # RGBA8 strip of this rank; with N > 1 two sets of buffers, so that the gather of frame f (asynchronous, on RCCL's
        # stream) overlaps the kernels of frame f + 1 and is only waited for before its buffers are written again
        pbos = [torch.zeros((strips.max_rows * WIDTH, 4), dtype=torch.uint8, device="cuda") for _ in range(2 if world > 1 else 1)]
        gather_outs = [[torch.empty_like(pbos[0]) for _ in range(world)] if (world > 1 and rank == 0) else None for _ in pbos]
        pending = [None] * len(pbos)

        def frame():
            move_camera(strips.looper)
            strips.frame(REUSE, 0)             # GBuffer::render, ReSTIRDirect (phase A, halo, phase B), GBuffer::update
            if sobol_num is not None:
                strips.looper %= sobol_num
            k = state["frame_no"] % len(pbos); state["frame_no"] += 1
            if pending[k] is not None:
                pending[k].wait(); pending[k] = None
            capi.copy_image_to_pbo(pbos[k].data_ptr(), backend.image.data_ptr() + y0 * WIDTH * 12, WIDTH, rows, TONEMAP, 1.0)
            if world > 1:
                pending[k] = dist.gather(pbos[k], gather_outs[k], dst=0, async_op=True)

        def finish_gathers():
            for k in range(len(pending)):
                if pending[k] is not None:
                    pending[k].wait(); pending[k] = None

        def pbo_ptr():
            return pbos[0].data_ptr()

    def barrier():
        finish_gathers()                   # every frame's image has reached rank 0 before the clock stops
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Before the warm-up: the library measures once per scene whether GBuffer::render is walked together with the primary rays
    # (frames 2-13 of a new rs_restir, restir.hip); those frames run here, so that warm-up and timed frames all use the form it
    # chose.  Strips too small for the fused launch have nothing to choose (-2).
    # The count is the same on every rank (a frame exchanges halo rows with the neighbours).  The library decides at the first frame
    # end after its last time stamp (frame 14) has been reached -- it never waits on the host -- hence the synchronisation and
    # the two frames after it.
    calibration_frames = 18
    for _ in range(calibration_frames - 2):
        frame()
    barrier()
    for _ in range(2):
        frame()
    barrier()
    for _ in range(args.warmup):
        frame()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    barrier()
    elapsed = time.perf_counter() - t0
    timed_form = backend.restir.last_launch()             # (fused, chains) of the timed frames' launches: read before any other mode runs
    counted = min(args.steps, 1024)
    # G-buffer rays: only the strip's own rows count (the +-5 halo rows a strip re-renders are overhead, not throughput)
    local_rays = backend.restir.ray_total(counted) / counted * args.steps + rows * WIDTH * args.steps

    # latency of one frame when nothing overlaps: the reference's own mode (every call synchronises, cudaUtil.h:15)
    capi.set_sync(True)
    sync_ms = []
    for _ in range(12):
        torch.cuda.synchronize()
        ts = time.perf_counter()
        frame()
        finish_gathers()
        torch.cuda.synchronize()
        sync_ms.append((time.perf_counter() - ts) * 1e3)
    capi.set_sync(False)
    barrier()

    # per-pass times of a few extra (untimed) frames, HIP events on the library's stream; the G-buffer render is kept on
    # that stream for these frames so that every pass is timed alone (in the timed region above it overlaps the
    # primary-ray and RIS kernels from the library's second stream)
    # how long this rank's library stream sat waiting for the neighbours' border rows (the part of the exchange the interior rows
    # of phase B did not hide), frames overlapped as in the timed region, read after each frame
    halo_wait = []
    if driver == "c" and world > 1:
        drv.enable_timing(True)
        for _ in range(20):
            frame()
            finish_gathers()
            torch.cuda.synchronize()
            halo_wait.append(drv.halo_wait_ms())
        drv.enable_timing(False)
        barrier()

    capi.set_side_stream(0)
    backend.restir.enable_timing(True)
    spatial_ms, pass_ms = [], np.zeros(4)
    for _ in range(20):
        frame()
        torch.cuda.synchronize()
        ms = backend.restir.pass_times()
        spatial_ms.append(ms[3]); pass_ms += np.array(ms)
    pass_ms /= len(spatial_ms)
    # the two passes outside ReSTIRDirect (the library enqueues on the null stream, which is torch's current stream here)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    gb_ms, pbo_ms = [], []
    for _ in range(10):
        ev[0].record(); backend.gbuffer_render(y0, y1); ev[1].record()
        ev[2].record(); capi.copy_image_to_pbo(pbo_ptr(), backend.image.data_ptr() + y0 * WIDTH * 12, WIDTH, rows, TONEMAP, 1.0); ev[3].record()
        torch.cuda.synchronize()
        gb_ms.append(ev[0].elapsed_time(ev[1])); pbo_ms.append(ev[2].elapsed_time(ev[3]))
    capi.set_side_stream(4)
    backend.restir.enable_timing(False)

    finish_gathers()
    # N > 1 (and BENCH_FORCE_STRIPS): the strips' image against a full frame.  Fresh reservoirs, G-buffers and display buffers on every rank
    # for the strip driver, a full-frame renderer of its own on rank 0, the same six frames through both -- the strips through the very
    # frame() of the timed region, launches overlapped, the display image gathered behind the next frame.  After every frame the ranks' rows
    # of the RADIANCE image are gathered on rank 0 as well (rs_strips_gather) and compared with the full frame's BIT FOR BIT; at the end the
    # last two DISPLAY images (RGBA8, the asynchronous gathers) are compared with the full frame's tone-mapped ones byte for byte.
    strips_parity = None
    if driver == "c":
        chk = HipBackend(capi, scene, cam, WIDTH, HEIGHT)
        chk_state = {"looper": 0, "frame_no": 0}
        chk_pbos = [torch.zeros((HEIGHT * WIDTH, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        ref = StripRenderer(HipBackend(capi, scene, cam, WIDTH, HEIGHT), 1, 0, HEIGHT) if rank == 0 else None
        ref_pbos = [torch.zeros((HEIGHT * WIDTH, 4), dtype=torch.uint8, device="cuda") for _ in range(2)] if rank == 0 else None
        differing, l1_sum, frames_checked = 0, 0.0, 6
        for f in range(frames_checked):
            frame(chk, chk_state, chk_pbos)
            drv.gather(chk.image.data_ptr(), 12, 0)
            if ref is not None:
                ref.looper = f
                ref.frame(REUSE, 0)
                capi.copy_image_to_pbo(ref_pbos[f % 2].data_ptr(), ref.b.image.data_ptr(), WIDTH, HEIGHT, TONEMAP, 1.0)
                capi.synchronize(); torch.cuda.synchronize()
                a, b = chk.image.view(torch.int32), ref.b.image.view(torch.int32)
                differing += int((a != b).any(dim=1).sum().item())
                l1_sum += float((chk.image - ref.b.image).abs().sum(dim=1).mean().item())
                if os.environ.get("BENCH_PARITY_ROWS", "0") == "1":          # which rows differ (debugging a failed check)
                    bad_rows = (a != b).any(dim=1).view(HEIGHT, WIDTH).any(dim=1).nonzero().flatten().tolist()
                    print("bench.py: strips parity, frame %d: %d rows differ: %s (strip bounds %s)" % (f, len(bad_rows), bad_rows[:40], bounds), file=sys.stderr, flush=True)
        finish_gathers()
        capi.synchronize(); torch.cuda.synchronize()
        display_differing = None
        if ref is not None:
            display_differing = sum(int((chk_pbos[k] != ref_pbos[k]).any(dim=1).sum().item()) for k in range(2))
        if world > 1:
            dist.barrier()
        strips_parity = {"frames": frames_checked, "pixels_per_frame": WIDTH * HEIGHT, "differing_pixels": differing, "mean_l1": l1_sum / frames_checked,
                         "display_frames": 2, "display_differing_pixels": display_differing,
                         "checker": "rank 0's own full-frame render of the same frames from fresh reservoirs (librestir_hip, which the N = 1 line pins to the "
                                    "oracle): radiance of the gathered strips bit for bit after every frame, the last two asynchronously gathered RGBA8 "
                                    "display images byte for byte; launches overlapped as in the timed region"}
    t = torch.tensor([elapsed, float(local_rays)], dtype=torch.float64, device=ctl_device)
    if world > 1:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, total_rays = float(tmax[0]), float(tsum[1])
    else:
        total_rays = float(local_rays)

    # HBM bandwidth as this box delivers it to a plain device-to-device copy (report the roofline fraction "of spec"
    # and "of measured copy"): 1 GiB read + 1 GiB written per copy, best of 10
    copy_gbs = None
    if rank == 0:
        src = torch.empty(1 << 28, dtype=torch.float32, device="cuda").fill_(1.0)
        dst = torch.empty_like(src)
        best = 1e9
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dst.copy_(src); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        copy_gbs = 2.0 * src.numel() * 4 / (best * 1e-3) / 1e9
        del src, dst

    if rank == 0:
        spatial_us = float(np.median(spatial_ms)) * 1e3
        overlapped_us = overlapped_kernel_us("k_spatial_shade")
        algo_bytes = ALGO_BYTES_PER_PIXEL * WIDTH * rows
        achieved = algo_bytes / (spatial_us * 1e-6) / 1e9
        out = {
            "metric": "Mrays/s (1920x1080 ReSTIR-DI, 32 candidates, spatiotemporal reuse)",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_frame_synchronous": float(np.median(sync_ms[2:])),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 3: procedural Sponza-class seed 1 (262144 triangles, 1024 emissive), 1920x1080, "
                                   "32 RIS candidates, spatiotemporal ReSTIR-DI; frame = GBuffer::render + ReSTIRDirect + copyImageToPBO + GBuffer::update",
                       "camera": "orbit" if args.orbit else "static",
                       "sampler": "default engine (thrust minstd_rand, SAMPLER_USE_SOBOL false: the reference's default)" if sobol_num is None else
                                  "Sobol (src/sampler.h:9-36) over the build's own %d x 200 table" % sobol_num,
                       "launches": "asynchronous (rs_set_sync(0)): consecutive frames overlap on the library's auxiliary streams; GBuffer::render is walked "
                                   "together with the primary rays when the library measures that to be faster (it measures in untimed frames before the warm-up); "
                                   "ms_per_frame_synchronous is one frame alone with a synchronisation after every call, the reference's mode",
                       "tiling": (f"{world} row strips of " + "/".join(str(b - a) for a, b in bounds) + " rows (" +
                                  ("equal heights" if os.environ.get("BENCH_EVEN_STRIPS", "0") == "1" else "cost-balanced by measurement") + "), "
                                  "5 border rows of reservoirs + G-buffer id / normal / depth point-to-point (68 B/px), RGBA8 gather to rank 0") if world > 1 else "none",
                       "strip_driver": {"c": "librestir_hip rs_strips_frame / rs_strips_gather_begin,_end (restir_amd/csrc/strips.hip)",
                                        "py": "restir_amd/tiling.py over torch.distributed (%s)" % backend_name, "none": "none (single GPU: rs_gbuffer_render + rs_restir_direct)"}[driver],
                       "transport": ("none" if driver == "none" else "torch.distributed " + backend_name if driver == "py" else
                                     "RCCL ncclSend / ncclRecv groups on the library stream, one group per frame (rs_comm_create_rccl_lib), ncclComm_t made by ncclCommInitRank from " + str(rccl.path)
                                     if transport == "rccl" else "host callbacks over torch.distributed gloo (rehearsal, not RCCL)"),
                       "rccl_ranks": (world if (driver == "c" and transport == "rccl") else (world if (driver == "py" and backend_name == "nccl" and world > 1) else 0)),
                       "strip_rows_per_rank": [b - a for a, b in bounds],
                       "strip_driver_fallback": fallback,
                       "halo_wait_ms_rank0": (float(np.median(halo_wait)) if halo_wait else None),
                       "rays_per_frame": total_rays / args.steps},
            "roofline": {"bound": "hbm", "kernel": "k_spatial_shade", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic("k_spatial_shade")[0] if world == 1 else None,
                         "traffic_source": "profiles/" + os.path.basename(PMC_SUMMARY) + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; "
                                           "the kernel lasted %s us there)" % pmc_traffic("k_spatial_shade")[1],
                         "algorithmic_bytes": algo_bytes, "kernel_us": spatial_us,
                         # the same kernel inside the timed region's mode shares the CUs with the kernels of the other frames: its duration
                         # there comes from the committed kernel trace of this command with the frames overlapped (events would need the
                         # host to wait inside the frames, which drains the overlap they are meant to observe)
                         "kernel_us_in_overlapped_frame": overlapped_us,
                         "frac_in_overlapped_frame": (algo_bytes / (overlapped_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if (overlapped_us and world == 1) else None,
                         "overlapped_source": "profiles/" + os.path.basename(OVERLAPPED_STATS),
                         "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs if copy_gbs else None,
                         "note": "rank 0's strip; the timed span includes the wait for the halo rows" if world > 1 else "full frame"},
            "pass_ms": {"gbuffer": float(np.median(gb_ms)), "to_rgba8": float(np.median(pbo_ms)), "primary": float(pass_ms[0]), "ris": float(pass_ms[1]), "shadow_temporal": float(pass_ms[2]), "spatial_shade": float(pass_ms[3])},
        }
        fused, chains = timed_form                            # what the timed frames launched, as the library reported it then
        form = "one fused launch" if fused == 1 else "two launches"
        how = {-2: "not measured: a launch below three rounds of wave slots", -1: "measurement not finished within this run", 0: "measured", 1: "measured"}[backend.restir.launch_choice()]
        out["config"]["launch_choice"] = "%s (%s), chains on %d streams in turn" % (form, how, chains)
        out["config"]["calibration_frames_before_warmup"] = calibration_frames
        if strips_parity is not None:
            out["parity" if world > 1 else "strips_parity"] = strips_parity       # N = 1 through the strip driver keeps the oracle's `parity` below
        if world == 1 and args.cpu_frames > 0 and sobol_num is None:
            out["cpu_baseline"], oracle_images = cpu_baseline(sd, args.cpu_frames)
            out["parity"] = parity_against(capi, sd, scene, oracle_images)
            out["cpu_config1"] = cpu_config1(out["cpu_baseline"]["cores"])
            ref_loop = cpu_reference_loop(sd, out["cpu_baseline"]["cores"])
            if ref_loop is not None:
                out["cpu_reference_loop"] = ref_loop
        print(json.dumps(out), flush=True)
    # orderly teardown: the strip driver (waits for its streams), then the communicator, then the control plane
    if driver == "c":
        capi.synchronize(); torch.cuda.synchronize()
        drv.destroy(); comm.destroy()
        if rccl is not None:
            if world > 1:
                dist.barrier()                 # every rank has left its last transfer before any rank destroys the communicator
            rccl.destroy()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
