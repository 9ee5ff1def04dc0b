#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: Mrays/s and ms/frame of the ReSTIR-DI per-frame sequence (runCuda,
src/main.cpp:146-185: GBuffer::render -> ReSTIRDirect -> [LeveledEAWFilter] -> copyImageToPBO -> GBuffer::update), 32 candidates,
spatiotemporal reuse, on the procedural scenes of BASELINE.json's configs.

    python bench.py --gpus N --steps K --warmup W [--config 3|4|5]

  --config 3 (default, the headline)  Sponza-class scene (262 144 triangles, 1 024 emissive), 1920x1080
  --config 4                          the same scene at 3840x2160 (the config BASELINE defines on 8 GPUs)
  --config 5                          Bistro-class scene (2.83 M triangles, 10 240 emissive), 1920x1080, + the 5-level EAW filter

A step = one frame.  A ray = one BVH walk (intersect or testOcclusion call): 1 G-buffer ray + 1 shading ray per pixel + 1 shadow ray
per shaded pixel; counted by the kernels.

N > 1: the framebuffer is cut into N row strips (strong scaling: the total work is fixed).  `python bench.py --gpus N` starts its N
rank processes itself (fresh children through torch.distributed.run, before this process has made any GPU call) and relays rank 0's
JSON line; started under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.  Every rank renders its strip, exchanges
5 border rows of published reservoirs and of the G-buffer id / normal / depth planes with its strip neighbours over RCCL
point-to-point between phase A and phase B, [filters its strip, exchanging the border rows of every level,] tone-maps its strip, and
the RGBA8 strips are gathered on rank 0 (asynchronously: the gather of one frame overlaps the next frame's kernels; the last gathers
are waited for before the clock stops) -- all inside the timed region.  The frames go through the PRODUCT's strip driver, the C ABI
of restir_amd/csrc/strips.hip (rs_strips_frame / rs_strips_eaw_filter / rs_strips_gather_begin / _end over rs_comm_create_rccl_lib),
on an ncclComm_t this script creates the way a C++ caller does (ncclGetUniqueId on rank 0, the id broadcast, ncclCommInitRank;
restir_amd/rccl.py); torch.distributed (gloo) is the control plane only: the id broadcast, barriers, the reduction of the timings.
  BENCH_STRIP_DRIVER=py      the Python form of the same schedule (restir_amd/tiling.py over torch.distributed) for cross-checks
  BENCH_STRIP_TRANSPORT=gloo the C driver over host callbacks + gloo: rehearsal on a one-GPU box (with BENCH_DEVICE=0 every rank
                             uses the same card; RCCL refuses two ranks on one device)
  BENCH_FORCE_STRIPS=1       N = 1 through the strip driver and a one-rank ncclComm as well
  BENCH_WATCHDOG=seconds     every rank (and the launcher) ends itself after that long (default 900 with N > 1)
  BENCH_COMM_STREAM=library|own|auto   where the strip driver enqueues its RCCL transfers (rs_strips_set_comm_stream); auto (default over
                             RCCL) times both over 40 frames before the warm-up and takes the own stream only if it wins by 3 %
  BENCH_STREAM_LEVEL=-1|0|1|2          rs_set_internal_stream_priority: the preferred level of the library's own streams (default 2 =
                             automatic; 1 with the denoiser of config 5).  The streams themselves are chosen by measurement
                             (`config.internal_streams`, per rank in `per_rank`), again after ncclCommInitRank

One JSON line is printed by rank 0; besides the contract's fields it carries
  roofline      the spatial-reuse pass (k_spatial_shade): algorithmic 92 B/px over its average launch duration by HIP
                events (20 back-to-back launches on the stream the kernel is launched on), against the 8 TB/s HBM3E peak; `traffic` is
                filled from profiles/ PMC runs when known
  roofline_eaw  config 5: the five a-trous level kernels, 44 B/px and level
  per_rank      N > 1: every rank's rows, ms per step, wait for the halo rows, per-pass times, its choice of internal streams
  expected_compute_only_ms   N > 1: the slowest rank's frame period of this split as measured on ONE GPU over a transport that moves nothing
                (tools/strip_period.py, profiles/r06_strip_period_c<config>.json): ms_per_step minus this is the wire, RCCL's launches, the ranks' skew
  ms_per_frame_single_in_flight   launches asynchronous, one frame at a time (between ms_per_step, three frames in flight, and
                ms_per_frame_synchronous, a synchronisation after every call)
  ms_per_step_sustained / sustained   --sustained-frames (default 2 000) more overlapped frames AFTER the timed region, in ten windows: their mean and
                the fastest / slowest window (the driver's --steps 20 is a 20-ms burst)
  failed / error / failing_rank / phase   only when the run failed or hung: the launcher's diagnosis (which rank, in which phase)
  cpu_reference_loop  closest-hit Mrays/s of a host loop over the reference's own compiled intersection code (primary rays only)
  cpu_baseline  the CPU oracle (oracle/restir_oracle.c, OpenMP) on this box's host cores, on a bounded sample of the same workload
                (rank 0, N = 1 only)
  parity        N = 1: the frames cpu_baseline rendered, rendered again by the GPU and compared bit for bit (differing_pixels, mean_l1)
                with the oracle's cos / sin correctly rounded (the mode the product is exact in);
                N > 1: six frames through the strip driver from fresh reservoirs, gathered on rank 0 and compared bit for bit with rank 0's
                own full-frame render of the same frames (`strips_parity` when BENCH_FORCE_STRIPS=1 sends N = 1 through the strip driver)
  parity_glibc  N = 1: the same GPU frames against the oracle in the mode the compiled reference pins (glibc's cosf / sinf): mean
                per-pixel L1, fraction of pixels beyond 1e-3, differing pixels -- the tolerance north_star states
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

REUSE = 3                      # ReservoirReuse::Spatiotemporal
TONEMAP = 2                    # ToneMapping::ACES (Settings default, src/common.cpp:4)
ALGO_BYTES_PER_PIXEL = 92      # spatial-reuse pass
EAW_BYTES_PER_PIXEL_LEVEL = 44 # (12 colour + 20 id/normal/depth) read + 12 written per pixel and level
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec

CONFIGS = {
    3: dict(width=1920, height=1080, scene="sponza", denoise=False, cpu_frames=3,
            workload="BASELINE config 3: procedural Sponza-class seed 1 (262144 triangles, 1024 emissive), 1920x1080, 32 RIS candidates, "
                     "spatiotemporal ReSTIR-DI; frame = GBuffer::render + ReSTIRDirect + copyImageToPBO + GBuffer::update"),
    4: dict(width=3840, height=2160, scene="sponza", denoise=False, cpu_frames=1,
            workload="BASELINE config 4: procedural Sponza-class seed 1 (262144 triangles, 1024 emissive), 3840x2160, 32 RIS candidates, "
                     "spatiotemporal ReSTIR-DI; frame = GBuffer::render + ReSTIRDirect + copyImageToPBO + GBuffer::update"),
    5: dict(width=1920, height=1080, scene="bistro", denoise=True, cpu_frames=2,
            workload="BASELINE config 5: procedural Bistro-class seed 2 (2830336 triangles, 10240 emissive), 1920x1080, 32 RIS candidates, "
                     "spatiotemporal ReSTIR-DI + LeveledEAWFilter (5 levels); frame = GBuffer::render + ReSTIRDirect + LeveledEAWFilter::filter + "
                     "copyImageToPBO + GBuffer::update"),
}


def profile_file(config, suffix):
    """The committed rocprofv3 summary of this command for `config` (newest round first; None if there is none)."""
    for name in ("r06_config%d_%s" % (config, suffix), "r05_config%d_%s" % (config, suffix), "r04_config%d_%s" % (config, suffix)) + (("r03_final2_%s" % suffix,) if config == 3 else ()):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            return p
    return None


def pmc_traffic(config, kernel):
    """(HBM bytes per launch, the kernel's average duration in that profile, file): FETCH_SIZE x 2 + WRITE_SIZE (the gfx950 correction of
    MI355X_MICROARCH.md) from the separate --pmc passes of tools/profile.sh on this same command, condensed by
    tools/summarize_profile.py.  Counters cannot be read from inside the process, so `roofline.traffic` quotes the committed summary
    (null if it is absent); the duration lets a reader see whether the summary still describes the kernel that ran here."""
    p = profile_file(config, "hbm_counters.json")
    try:
        with open(p) as fh:
            k = json.load(fh)["kernels"][kernel]
            return float(k["hbm_bytes_per_launch"]), (float(k["average_ns"]) / 1e3 if k.get("average_ns") else None), p
    except (OSError, KeyError, ValueError, TypeError):
        return None, None, p


def trace_kernel_us(config, kernel, suffix):
    """Average duration of a kernel in a committed rocprofv3 kernel trace of this command: suffix "kernel_stats.csv" = every kernel alone on
    one stream (RS_SIDE_STREAM=0), "kernel_stats_overlapped.csv" = `bench.py --only-timed`, whose every frame is an overlapped one (since
    round 6; the measurement's own launches of the spatial pass carry another name, k_spatial_shade_probe)."""
    p = profile_file(config, suffix)
    try:
        import csv
        with open(p) as fh:
            rows = [r for r in csv.reader(fh) if r and not r[0].startswith("#")]
        head = rows[0]
        for r in rows[1:]:
            if r[0] == kernel:
                return float(r[head.index("average_ns")]) / 1e3, p
    except (OSError, ValueError, IndexError, TypeError):
        pass
    return None, p


def overlapped_kernel_us(config, kernel):
    return trace_kernel_us(config, kernel, "kernel_stats_overlapped.csv")


def expected_compute_only(config, world):
    """The committed compute-only frame period of an N-way split of `config` (tools/strip_period.py: every rank of the split alone on ONE
    MI355X through rs_strips_frame over a transport that moves nothing, cost-balanced heights; profiles/r06_strip_period_c<config>.json, round 5's if that is absent):
    what the first multi-GPU run is to be compared with -- ms_per_step minus this is the wire, RCCL's launches and the ranks' skew."""
    p = os.path.join(ROOT, "profiles", "r06_strip_period_c%d.json" % config)
    if not os.path.exists(p):
        p = os.path.join(ROOT, "profiles", "r05_strip_period_c%d.json" % config)
    try:
        with open(p) as fh:
            t = json.load(fh)["worlds"]
        w, one = t[str(world)], t["1"]
        return {"ms": w["max_ms"], "per_rank_ms": w["ms"], "rows": w["rows"], "n1_ms_same_box": one["max_ms"],
                "speedup_compute_only": one["max_ms"] / w["max_ms"], "source": "profiles/" + os.path.basename(p)}
    except (OSError, KeyError, ValueError):
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


# ---- what a rank is doing, for the launcher's diagnosis of a failed run --------------------------------------------------------------------

PHASES = ("start", "calibrate", "comm_init", "first_frames", "comm_stream_trial", "warmup", "timed", "post_timing", "parity", "done")


def set_phase(name, **data):
    """Every rank of an N > 1 run records the phase it has entered (and, once it has them, its own timings) in
    $BENCH_STATUS_DIR/rank<r>.json -- written to a temporary name and renamed, so the launcher never reads half a file.  If the job dies
    or hangs, the launcher's JSON line says which rank was where (launch_ranks).  No-op without the variable (N = 1, tests of other parts)."""
    d = os.environ.get("BENCH_STATUS_DIR")
    if not d:
        return
    rank = int(os.environ.get("RANK", "0"))
    rec = {"rank": rank, "phase": name, "time": time.time()}
    rec.update(data)
    tmp = os.path.join(d, "rank%d.json.tmp" % rank)
    try:
        with open(tmp, "w") as fh:
            json.dump(rec, fh)
        os.replace(tmp, os.path.join(d, "rank%d.json" % rank))
    except OSError:
        pass


def read_status(d, n):
    """{rank: record} of the status files the ranks left in d (ranks that never got to write one are absent)."""
    out = {}
    for r in range(n):
        try:
            with open(os.path.join(d, "rank%d.json" % r)) as fh:
                out[r] = json.load(fh)
        except (OSError, ValueError):
            pass
    return out


def diagnose(n, rc, status, stderr_tail):
    """Which rank failed and where: the rank torch.distributed.run names as the first failure (its summary on stderr), else the rank that
    got least far (a hang: every rank that is not done waits for that one); its phase from its status file."""
    import re
    text = "".join(stderr_tail)
    failing = None
    m = re.search(r"Root Cause.*?rank\s*:\s*(\d+)\s*\(local_rank", text, re.S) or re.search(r"rank\s*:\s*(\d+)\s*\(local_rank", text)
    if m:
        failing = int(m.group(1))
    order = {p: i for i, p in enumerate(PHASES)}
    seen = {r: order.get(rec.get("phase"), -1) for r, rec in status.items()}
    for r in range(n):
        seen.setdefault(r, -1)                                 # never wrote a status file: died before "start"
    if failing is None:
        unfinished = [r for r in sorted(seen) if seen[r] < order["done"]]
        failing = min(unfinished, key=lambda r: seen[r]) if unfinished else None
    phase = status.get(failing, {}).get("phase", "before start") if failing is not None else None
    what = "the ranks were still running at the watchdog's limit" if rc == 124 else "the ranks ended with rc %d" % rc
    return {"failed": True, "rc": rc, "error": "%s; rank %s was in phase %s" % (what, failing, phase), "failing_rank": failing, "phase": phase,
            "phases": {str(r): status.get(r, {}).get("phase", "before start") for r in range(n)},
            "per_rank": [status[r]["mine"] for r in sorted(status) if status[r].get("mine")] or None,
            "stderr_tail": [l.rstrip() for l in list(stderr_tail)[-12:]]}


# ---- the launcher: `python bench.py --gpus N` with N > 1 and no WORLD_SIZE -------------------------------------------------------------

def launch_ranks(n, argv, script=None, watchdog=960.0, out=None, err=None):
    """Start the n rank processes of one node as fresh children -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node n
    --master-addr 127.0.0.1 --master-port <free> script argv...`, the command the driver itself uses -- from a process that has made
    no GPU call (nothing is re-exec'd).  Relays the ranks' stderr as it comes, keeps its tail, and prints ONE JSON line on stdout: the
    line rank 0 wrote, or -- when the job failed, hung or ended without one -- a line with "failed": true, "error", the failing rank, the
    phase it was in and whatever per-rank data the ranks had recorded (set_phase); a line rank 0 printed before ANOTHER rank failed is
    relayed with those fields added, so that a harness which parses stdout and ignores the exit code cannot take it for a result.
    Returns the exit code: the children's if they failed, 124 if they were still running after `watchdog` seconds (their process group
    -- the one started here -- is killed), 1 if they ended cleanly without a JSON line."""
    import collections
    import shutil
    import signal
    import socket
    import subprocess
    import tempfile
    import threading
    out = out or sys.stdout
    err = err or sys.stderr
    script = script or os.path.abspath(__file__)
    with socket.socket() as s:                              # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between processes on this driver
    status_dir = tempfile.mkdtemp(prefix="bench_status_")
    env["BENCH_STATUS_DIR"] = status_dir
    print("bench.py: starting %d ranks: %s" % (n, " ".join(cmd)), file=err, flush=True)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True, start_new_session=True)
    tail = collections.deque(maxlen=60)
    lines = []

    def pump_err():
        for line in p.stderr:
            tail.append(line)
            err.write(line); err.flush()

    def pump_out():
        for line in p.stdout:
            lines.append(line)

    threads = [threading.Thread(target=pump_err, daemon=True), threading.Thread(target=pump_out, daemon=True)]
    for t in threads:
        t.start()
    try:
        rc = p.wait(timeout=watchdog)
    except subprocess.TimeoutExpired:
        print("bench.py: the ranks did not finish within %.0f s: killing their process group" % watchdog, file=err, flush=True)
        try:
            os.killpg(p.pid, signal.SIGKILL)               # the session started above: the launcher and its ranks, nothing else
        except ProcessLookupError:
            pass
        p.wait()
        rc = 124
    for t in threads:
        t.join(timeout=10)
    result = None
    for line in lines:
        s = line.strip()
        if s.startswith("{") and '"metric"' in s:
            try:
                json.loads(s)
                result = s
                continue
            except ValueError:
                pass
        if s:
            err.write("[ranks stdout] " + line)
    if rc == 0 and result is None:
        print("bench.py: the ranks ended with rc 0 but printed no JSON line", file=err, flush=True)
        rc = 1
    if rc != 0:
        print("bench.py: the ranks ended with rc %d; last lines of their stderr:\n%s" % (rc, "".join(list(tail)[-25:])), file=err, flush=True)
        diag = diagnose(n, rc, read_status(status_dir, n), tail)
        rec = json.loads(result) if result is not None else {"metric": "Mrays/s", "value": None, "unit": "Mrays/s", "n_gpus": n}
        if result is not None and diag.get("per_rank") is None:
            diag.pop("per_rank")                            # keep the line's own
        rec.update(diag)
        result = json.dumps(rec)
        print("bench.py: " + diag["error"], file=err, flush=True)
    shutil.rmtree(status_dir, ignore_errors=True)
    out.write(result + "\n"); out.flush()
    return rc


# ---- the checker's side (after the timed region; rank 0, N = 1) ---------------------------------------------------------------------------

def oracle_frames(cfg, sd, frames, libm_mode, timed):
    """`frames` + 1 frames of the workload on the oracle (looper 0.., static camera, fresh reservoirs): the images a GPU run of the same
    frames is compared with; with `timed`, the cpu_baseline record (the first frame untimed: thread-pool start-up, page faults)."""
    import numpy as np
    W, H = cfg["width"], cfg["height"]
    threads = host_threads()
    os.environ["OMP_NUM_THREADS"] = str(threads)          # read by libgomp when liboracle.so is loaded
    from oracle import binding as ob
    from tests.common import OracleRenderer
    ob.set_libm_mode(libm_mode)
    try:
        o = OracleRenderer(sd, W, H)
        images, filtered = [], []
        rays, dt = 0, 0.0
        for f in range(frames + 1):
            t0 = time.perf_counter()
            o.gbuf.render(o.scene, o.cam)
            o.rays = o.restir.direct(o.scene, o.cam, o.gbuf, o.image, 0, o.looper, REUSE)
            if cfg["denoise"]:
                filtered.append(ob.eaw_filter(o.gbuf, o.cam, o.image).copy())
            o.gbuf.update(o.cam)
            o.looper += 1
            if f > 0:
                dt += time.perf_counter() - t0
                rays += o.rays + W * H
            images.append(o.image.copy())
    finally:
        ob.set_libm_mode(0)
    rec = None
    if timed and frames > 0:
        rec = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "ms_per_frame": dt / frames * 1e3,
               "sample": f"{frames} frames of the same workload ({W}x{H} spatiotemporal, {cfg['scene']}-class scene"
                         + (", 5-level EAW filter" if cfg["denoise"] else "") + "), OpenMP over rows"}
    return rec, images, filtered


def gpu_frames(capi, cfg, sd, scene, count):
    """The checker's frames once more on the GPU, from fresh reservoirs, in the reference's synchronous mode (host copies)."""
    import torch
    from tests.common import HipRenderer
    W, H = cfg["width"], cfg["height"]
    capi.set_sync(True)
    h = HipRenderer(capi, sd, W, H, scene=scene)
    eaw = capi.EAWFilter(W, H, 5) if cfg["denoise"] else None
    out = torch.zeros_like(h.image) if eaw else None
    images, filtered = [], []
    for f in range(count):
        h.gbuf.render(h.scene, h.cam)
        h.restir.direct(h.scene, h.cam, h.gbuf, h.image.data_ptr(), 0, f, REUSE)
        if eaw:
            p = eaw.filter(out.data_ptr(), h.image.data_ptr(), h.gbuf, h.cam)
            capi.synchronize()
            res = torch.empty_like(h.image)
            capi.hip_memcpy_d2d(res.data_ptr(), p, res.numel() * 4)
            filtered.append(res.cpu().numpy())
        h.gbuf.update(h.cam)
        images.append(h.image.cpu().numpy())
    if eaw:
        eaw.destroy()
    capi.set_sync(False)
    return images, filtered


def compare_frames(cfg, ref_images, got_images, ref_filtered, got_filtered, checker):
    import numpy as np
    differing, l1, flips, worst = 0, 0.0, 0.0, 0.0
    for ref, got in zip(ref_images, got_images):
        d = np.abs(ref.astype(np.float64) - got.astype(np.float64)).sum(axis=1)
        differing += int(np.count_nonzero((ref.view(np.uint32) != got.view(np.uint32)).any(axis=1)))
        l1 += float(d.mean()); flips += float(np.mean(d > 1e-3)); worst = max(worst, float(d.max()))
    n = len(ref_images)
    rec = {"frames": n, "pixels_per_frame": cfg["width"] * cfg["height"], "differing_pixels": differing, "mean_l1": l1 / n,
           "fraction_beyond_1e-3": flips / n, "max_l1": worst, "checker": checker}
    if ref_filtered:
        # the filter's exponentials are the hardware's: stated tolerance rtol 1e-5 (tests/test_gpu_parity.py test_eaw_filter)
        err = max(float(np.max(np.abs(a.astype(np.float64) - b) / (1e-6 + 1e-5 * np.abs(a.astype(np.float64))))) for a, b in zip(ref_filtered, got_filtered))
        rec["eaw_error_over_tolerance"] = err              # <= 1 : within atol 1e-6 + rtol 1e-5 everywhere
    return rec


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


def cpu_reference_loop(cfg, sd, threads):
    """The baseline north_star names: a host-side loop over the reference's own intersections.h / bvh.h -- DevScene::intersect's
    traversal around the reference's compiled AABB::intersect / intersectTriangle on the reference builder's tree
    (oracle/_ref/libref_subset.so, prebuilt in the build container; oracle/ref_subset.cpp).  Closest hits of the benchmark view's
    camera rays (one jittered ray per pixel of the frame), repeated for about five seconds.  None when the library is not there."""
    import ctypes as C
    import numpy as np
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle import binding as ob
    R = ob.ref_subset()
    if R is None:
        return None
    w, h = cfg["width"], cfg["height"]
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
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json config (3 = the headline)")
    ap.add_argument("--cpu-frames", type=int, default=None, help="frames timed for cpu_baseline (0 = skip; default depends on the config)")
    ap.add_argument("--orbit", action="store_true", help="orbit the camera (runCuda animateCamera) instead of the static default")
    ap.add_argument("--sustained-frames", type=int, default=2000, help="overlapped frames run AFTER the timed region for ms_per_step_sustained (10 windows; 0 = skip)")
    ap.add_argument("--only-timed", action="store_true", help="stop after the timed region and the sustained frames (profiling: every frame of the process is then an overlapped one)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    WIDTH, HEIGHT, DENOISE = cfg["width"], cfg["height"], cfg["denoise"]
    if args.cpu_frames is None:
        args.cpu_frames = cfg["cpu_frames"]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's own form of the command: this process becomes the launcher (no GPU call has been made, and none will be)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], watchdog=float(os.environ.get("BENCH_WATCHDOG", "900")) + 60.0))

    # A rank that waits for a peer that will never answer would hang the whole job: with N > 1 every rank ends itself (Python
    # stacks of all threads on stderr) if the run takes longer than BENCH_WATCHDOG seconds (default 900; N = 1: only when set).
    watchdog = os.environ.get("BENCH_WATCHDOG", "900" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "")
    if watchdog:
        import faulthandler
        faulthandler.dump_traceback_later(float(watchdog), exit=True)

    set_phase("start")
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU (plain `python bench.py --gpus N` starts the ranks itself)")
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

    sd = scenes.sponza_class(seed=1, scale=1.0) if cfg["scene"] == "sponza" else scenes.bistro_class(seed=2, scale=1.0)
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
    # a frame that runs the denoiser on the library stream does better with the library's own streams BELOW the caller's
    # (rs_set_internal_stream_priority, before the first frame; profiles/r05_ab_stream_levels_by_workload.log)
    capi.set_internal_stream_priority(int(os.environ.get("BENCH_STREAM_LEVEL", "1" if DENOISE else "2")))
    # BENCH_DENOISE_STREAM=1: the filter, the tone map of its result and the display gather on the library's denoise stream
    # (rs_set_denoise_stream; measured slower on config 5, profiles/r06_ab_denoise_stream_mode1.log: default 0)
    capi.set_denoise_stream(int(os.environ.get("BENCH_DENOISE_STREAM", "0")) if DENOISE else 0)
    capi.prepare_streams()                 # the library chooses its own streams NOW (about 25 ms), not inside the first frame
    min_rows = 32 if DENOISE else 8        # the EAW levels on strips reach 32 rows (rs_strips_eaw_filter)
    # N > 1: strip heights balanced by measured cost before the warm-up (rows near the horizon cost several times a sky row and
    # the slowest strip sets the frame time); BENCH_EVEN_STRIPS=1 keeps equal heights
    bounds = None
    set_phase("calibrate")
    if world > 1 and os.environ.get("BENCH_EVEN_STRIPS", "0") != "1":
        bounds = calibrate_bounds(backend, world, rank, HEIGHT, dist, torch.cuda.synchronize, reuse=REUSE, denoise=DENOISE, min_rows=min_rows)
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

    def next_looper(looper):
        return looper + 1 if sobol_num is None else (looper + 1) % sobol_num

    rccl = None
    fallback = None
    if driver == "c":
        # ---- the product's strip driver: strips.hip through the C ABI ---------------------------------------------------------
        from restir_amd.rccl import GlooTransport, RcclComm
        set_phase("comm_init")
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
        # ncclCommInitRank made streams of its own after the library had chosen its streams (the calibration of the strip heights ran
        # frames): which streams run side by side depends on every stream of the process, so the library chooses again
        capi.choose_internal_streams_again()
        capi.prepare_streams()
        drv = capi.Strips(comm, WIDTH, HEIGHT, [b[0] for b in bounds] + [HEIGHT])
        if DENOISE and world > 1 and os.environ.get("BENCH_GBUFFER_HALO", "32") != "5":
            drv.set_gbuffer_halo(32)           # the 32 G-buffer rows the filter's taps reach travel with the reservoir rows: one exchange per frame less
        y0, y1 = drv.y0, drv.y1
        assert (y0, y1) == tuple(bounds[rank])
        rows = y1 - y0
        # the display image, full-frame sized on every rank (rs_strips_gather addresses rows in place); two of them, so that the
        # gather of frame f travels while frame f + 1 renders
        pbos = [torch.zeros((HEIGHT * WIDTH, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        eaw = capi.EAWFilter(WIDTH, HEIGHT, 5) if DENOISE else None

        def frame(bk=None, st=None, pb=None):
            # (bk, st, pb: another set of render objects, frame counters and display buffers -- the parity check below runs this very
            # function on fresh ones)
            bk, st, pb = bk or backend, st or state, pb or pbos
            move_camera(st["looper"])
            drv.frame(bk.restir, scene, cam, bk.gbuf, bk.image.data_ptr(), 0, st["looper"], REUSE)
            shown = bk.image.data_ptr()
            if DENOISE:
                shown = drv.eaw_filter(eaw, bk.gbuf, cam, bk.image.data_ptr())      # rows [y0, y1) of the driver's result buffer
            st["shown"] = shown
            bk.gbuf.update(cam)
            if args.orbit and world > 1:
                drv.exchange_history(bk.restir, bk.gbuf)
            st["looper"] = next_looper(st["looper"])
            k = st["frame_no"] % 2; st["frame_no"] += 1
            drv.gather_end(k)                                           # the gather that read pb[k] two frames ago
            capi.copy_image_to_pbo(pb[k].data_ptr() + y0 * WIDTH * 4, shown + y0 * WIDTH * 12, WIDTH, rows, TONEMAP, 1.0)
            drv.gather_begin(pb[k].data_ptr(), 4, 0, k)

        def finish_gathers():
            drv.gather_end(0); drv.gather_end(1)

        def pbo_ptr():
            return pbos[0].data_ptr() + y0 * WIDTH * 4
    else:
        # ---- restir_amd/tiling.py over torch.distributed (N = 1: the plain calls) -----------------------------------------------
        # (init_process_group / ncclCommInitRank made streams after the library had chosen its own during the calibration of the strip
        # heights: which streams run side by side depends on every stream of the process, so it chooses again -- as the C driver's branch does)
        if world > 1:
            capi.choose_internal_streams_again()
            capi.prepare_streams()
        strips = StripRenderer(backend, world, rank, HEIGHT, dist=dist if world > 1 else None, share_history=args.orbit, bounds=bounds)
        y0, y1 = strips.y0, strips.y1
        rows = y1 - y0
        # RGBA8 strip of this rank; with N > 1 two sets of buffers, so that the gather of frame f (asynchronous, on RCCL's
        # stream) overlaps the kernels of frame f + 1 and is only waited for before its buffers are written again
        pbos = [torch.zeros((strips.max_rows * WIDTH, 4), dtype=torch.uint8, device="cuda") for _ in range(2 if world > 1 else 1)]
        gather_outs = [[torch.empty_like(pbos[0]) for _ in range(world)] if (world > 1 and rank == 0) else None for _ in pbos]
        pending = [None] * len(pbos)
        eaw = capi.EAWFilter(WIDTH, HEIGHT, 5) if (DENOISE and world == 1) else None
        eaw_out = torch.zeros_like(backend.image) if eaw else None

        def frame():
            move_camera(strips.looper)
            if eaw is not None:                # N = 1 with the filter: the plain calls, LeveledEAWFilter::filter as one call (rs_eaw_filter)
                backend.gbuffer_render(0, HEIGHT)
                backend.phase_a(strips.looper, REUSE, 0, HEIGHT)
                backend.phase_b(0, REUSE, 0, HEIGHT)
                shown = eaw.filter(eaw_out.data_ptr(), backend.image.data_ptr(), backend.gbuf, cam)
                backend.end_frame()
                strips.looper += 1
            else:
                strips.frame(REUSE, 0, denoise=DENOISE)    # GBuffer::render, ReSTIRDirect (phase A, halo, phase B), [filter], GBuffer::update
                shown = (strips.filtered if DENOISE else backend.image).data_ptr()
            state["shown"] = shown
            if sobol_num is not None:
                strips.looper %= sobol_num
            k = state["frame_no"] % len(pbos); state["frame_no"] += 1
            if pending[k] is not None:
                pending[k].wait(); pending[k] = None
            capi.copy_image_to_pbo(pbos[k].data_ptr(), shown + y0 * WIDTH * 12, WIDTH, rows, TONEMAP, 1.0)
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
    # (frames 2-33 of a new rs_restir, restir.hip); those frames run here, so that warm-up and timed frames all use the form it
    # chose.  Strips too small for the fused launch have nothing to choose (-2).
    # The count is the same on every rank (a frame exchanges halo rows with the neighbours).  The library decides at the first frame
    # end after its last time stamp (frame 34) has been reached -- it never waits on the host -- hence the synchronisation and
    # the two frames after it.
    calibration_frames = 38
    set_phase("first_frames")
    for _ in range(calibration_frames - 2):
        frame()
    barrier()
    for _ in range(2):
        frame()
    barrier()
    # N > 1 over RCCL: where the strip driver enqueues its transfers -- on the library stream (one group per frame, nothing overlaps the
    # wire) or on a stream of its own (the interior rows of phase B run while the border rows travel, at the price of a fifth stream) -- is
    # a property of the machine that no one-GPU box can measure: both are timed here, before the warm-up, and the faster one (by the slowest
    # rank) runs the warm-up and the timed frames.  Results do not depend on it (strips_loopback_ranks compares both with the full frame).
    # BENCH_COMM_STREAM=library|own skips the measurement.
    comm_choice, comm_trial = None, None
    set_phase("comm_stream_trial")
    if driver == "c" and world > 1:
        want = os.environ.get("BENCH_COMM_STREAM", "auto" if transport == "rccl" else "library")
        if want == "auto":
            # The library stream stays unless the driver's own stream wins by more than 3 % over 40 frames each (a single short trial
            # flips on noise, and the own-stream form has only ever carried data over the in-process loopback transport); both times go
            # into the JSON line.  To pin the choice for a scaling curve: BENCH_COMM_STREAM=library|own.
            trial = {}
            for own in (False, True):
                barrier()
                drv.set_comm_stream(own)
                for _ in range(4):
                    frame()
                barrier()
                ts = time.perf_counter()
                for _ in range(40):
                    frame()
                barrier()
                tt = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device=ctl_device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                trial[own] = float(tt[0]) / 40 * 1e3
            own = trial[True] < 0.97 * trial[False]
            comm_choice = "%s (measured over 40 frames each: library stream %.4f ms/frame, own stream %.4f; the own stream is taken when it wins by more than 3 %%)" % (
                "own stream" if own else "library stream", trial[False], trial[True])
            comm_trial = {"library_stream_ms": trial[False], "own_stream_ms": trial[True]}
        else:
            own = want == "own"
            comm_choice = ("own stream" if own else "library stream") + " (set by BENCH_COMM_STREAM)"
        barrier()
        drv.set_comm_stream(own)
    set_phase("warmup")
    for _ in range(args.warmup):
        frame()
    barrier()
    set_phase("timed")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    barrier()
    elapsed = time.perf_counter() - t0
    set_phase("post_timing", ms_per_step=elapsed / args.steps * 1e3)
    own_elapsed = elapsed
    timed_form = backend.restir.last_launch()             # (fused, chains) of the timed frames' launches: read before any other mode runs
    counted = min(args.steps, 1024)
    # G-buffer rays: only the strip's own rows count (the +-5 halo rows a strip re-renders are overhead, not throughput)
    local_rays = backend.restir.ray_total(counted) / counted * args.steps + rows * WIDTH * args.steps

    # A sustained figure next to the timed region's (the driver's --steps 20 is a burst of ~20 ms): --sustained-frames more overlapped
    # frames in ten windows, each bracketed like the timed region (the windows' ends drain the pipeline: ten drains in all).
    sustained = None
    if args.sustained_frames > 0:
        per = max(1, args.sustained_frames // 10)
        win = []
        for _ in range(10):
            ts = time.perf_counter()
            for _ in range(per):
                frame()
            barrier()
            tt = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device=ctl_device)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            win.append(float(tt[0]) / per * 1e3)
        sustained = {"frames": per * 10, "windows": 10, "ms_per_step": sum(win) / len(win), "min_window_ms": min(win), "max_window_ms": max(win)}
    # the spatial pass INSIDE overlapped frames, live: two events around the pass of each of 64 more frames, kept in a ring by the library
    # and read after the last one (rs_restir_enable_timing(r, 2) / rs_restir_spatial_times: nothing waits inside the frames)
    backend.restir.enable_timing(2)
    for _ in range(64):
        frame()
    barrier()
    spatial_in_frame_ms = backend.restir.spatial_times(64)
    backend.restir.enable_timing(False)
    if args.only_timed:
        if rank == 0:
            print(json.dumps({"metric": "Mrays/s", "only_timed": True, "n_gpus": world, "steps": args.steps, "ms_per_step": elapsed / args.steps * 1e3,
                              "sustained": sustained, "spatial_in_frame_us": float(np.mean(spatial_in_frame_ms)) * 1e3 if spatial_in_frame_ms else None}), flush=True)
        finish_gathers()
        capi.synchronize(); torch.cuda.synchronize()
        if driver == "c":
            drv.destroy(); comm.destroy()
            if rccl is not None:
                if world > 1:
                    dist.barrier()
                rccl.destroy()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        set_phase("done")
        return

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

    # one frame in flight, launches asynchronous: the latency an interactive loop sees when it does not pipeline frames ahead of its input
    # (src/preview.cpp:337-361) but lets the library overlap what it can WITHIN the frame (the render next to -- or fused with -- the primary
    # rays, RIS and shadow rays behind them); between the overlapped figure (three frames in flight) and the synchronous one
    single_ms = []
    for _ in range(14):
        torch.cuda.synchronize()
        ts = time.perf_counter()
        frame()
        finish_gathers()
        capi.synchronize(); torch.cuda.synchronize()
        single_ms.append((time.perf_counter() - ts) * 1e3)
    barrier()

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

    # per-pass times of a few extra (untimed) frames, HIP events on the library's stream; the G-buffer render is kept on
    # that stream for these frames so that every pass is timed alone (in the timed region above it overlaps the
    # primary-ray and RIS kernels from the library's second stream)
    capi.set_side_stream(0)
    backend.restir.enable_timing(True)
    spatial_ms, pass_ms = [], np.zeros(4)
    for _ in range(20):
        frame()
        torch.cuda.synchronize()
        ms = backend.restir.pass_times()
        spatial_ms.append(ms[3]); pass_ms += np.array(ms)
    pass_ms /= len(spatial_ms)
    # The spatial pass again, twenty launches back to back between two events: a single launch between two events carries 2-4 us of
    # launch gap and event latency that are not the kernel's (47-49 us by single-launch events against 45.8 in the rocprofv3 trace);
    # the average over back-to-back launches is what `roofline` quotes.  One more frame's render and phase A first, so that the pass
    # reads this frame's planes; with iter = 0 it is idempotent (it stores no reservoirs, restir.cu:188,211-212).
    backend.gbuffer_render(y0, y1)
    backend.phase_a(77, REUSE, y0, y1)                 # (any frame number: only the duration of the pass is read)
    backend.phase_b(0, REUSE, y0, y1)
    eb = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    backend.restir.set_probe(True)                     # these launches go out as k_spatial_shade_probe: a kernel trace tells them from the frames'
    torch.cuda.synchronize()
    eb[0].record()
    for _ in range(20):
        backend.phase_b(0, REUSE, y0, y1)
    eb[1].record()
    torch.cuda.synchronize()
    backend.restir.set_probe(False)
    spatial_b2b_us = eb[0].elapsed_time(eb[1]) / 20 * 1e3
    backend.end_frame()
    # the passes outside ReSTIRDirect (the library enqueues on the stream it was handed, which is torch's current stream here)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    gb_ms, pbo_ms = [], []
    for _ in range(10):
        ev[0].record(); backend.gbuffer_render(y0, y1); ev[1].record()
        ev[2].record(); capi.copy_image_to_pbo(pbo_ptr(), backend.image.data_ptr() + y0 * WIDTH * 12, WIDTH, rows, TONEMAP, 1.0); ev[3].record()
        torch.cuda.synchronize()
        gb_ms.append(ev[0].elapsed_time(ev[1])); pbo_ms.append(ev[2].elapsed_time(ev[3]))
    # config 5: the filter's position pass and its five levels, each alone (rs_eaw_positions_rows / rs_eaw_level_rows on this rank's rows;
    # the rows outside the strip hold what the last frame's exchange left there -- same work, the values do not matter here)
    eaw_level_ms = None
    if DENOISE:
        f = capi.EAWFilter(WIDTH, HEIGHT, 5)
        bufs = [torch.zeros_like(backend.image), torch.zeros_like(backend.image)]
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        acc = np.zeros(6)
        reps = 10
        for _ in range(reps + 2):
            evs[0].record()
            f.positions_rows(backend.gbuf, cam, max(0, y0 - 32), min(HEIGHT, y1 + 32))
            evs[1].record()
            for level in range(5):
                src = backend.image if level == 0 else bufs[(level - 1) % 2]
                f.level_rows(bufs[level % 2].data_ptr(), src.data_ptr(), backend.gbuf, level, y0, y1)
                evs[level + 2].record()
            torch.cuda.synchronize()
            if _ >= 2:
                acc += np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(6)])
        eaw_level_ms = (acc / reps).tolist()
        f.destroy()
        del bufs
    capi.set_side_stream(4)
    backend.restir.enable_timing(False)

    finish_gathers()
    # N > 1 (and BENCH_FORCE_STRIPS): the strips' image against a full frame.  Fresh reservoirs, G-buffers and display buffers on every rank
    # for the strip driver, a full-frame renderer of its own on rank 0, the same six frames through both -- the strips through the very
    # frame() of the timed region, launches overlapped, the display image gathered behind the next frame.  After every frame the ranks' rows
    # of the RADIANCE image (config 5: of the FILTERED image) are gathered on rank 0 as well (rs_strips_gather) and compared with the full
    # frame's BIT FOR BIT; at the end the last two DISPLAY images (RGBA8, the asynchronous gathers) are compared with the full frame's
    # tone-mapped ones byte for byte.
    strips_parity = None
    set_phase("parity", ms_per_step=own_elapsed / args.steps * 1e3)
    if driver == "c":
        chk = HipBackend(capi, scene, cam, WIDTH, HEIGHT)
        chk_state = {"looper": 0, "frame_no": 0}
        chk_pbos = [torch.zeros((HEIGHT * WIDTH, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        ref = StripRenderer(HipBackend(capi, scene, cam, WIDTH, HEIGHT), 1, 0, HEIGHT) if rank == 0 else None
        ref_pbos = [torch.zeros((HEIGHT * WIDTH, 4), dtype=torch.uint8, device="cuda") for _ in range(2)] if rank == 0 else None
        gathered = torch.zeros_like(chk.image)
        differing, l1_sum, frames_checked = 0, 0.0, 6
        for f in range(frames_checked):
            frame(chk, chk_state, chk_pbos)
            # the strip's rows of what was shown, copied into a buffer of our own (the driver's filter result buffer is the driver's); a
            # copy of the caller's own: with BENCH_DENOISE_STREAM=1 it needs the join first
            capi.join_denoise_stream(); capi.synchronize()
            capi.hip_memcpy_d2d(gathered.data_ptr() + y0 * WIDTH * 12, chk_state["shown"] + y0 * WIDTH * 12, rows * WIDTH * 12)
            drv.gather(gathered.data_ptr(), 12, 0)
            if ref is not None:
                ref.looper = f
                ref.frame(REUSE, 0, denoise=DENOISE)
                ref_img = ref.filtered if DENOISE else ref.b.image
                capi.copy_image_to_pbo(ref_pbos[f % 2].data_ptr(), ref_img.data_ptr(), WIDTH, HEIGHT, TONEMAP, 1.0)
                capi.synchronize(); torch.cuda.synchronize()
                a, b = gathered.view(torch.int32), ref_img.view(torch.int32)
                differing += int((a != b).any(dim=1).sum().item())
                l1_sum += float((gathered - ref_img).abs().sum(dim=1).mean().item())
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
                                    "oracle): " + ("filtered image" if DENOISE else "radiance") + " of the gathered strips bit for bit after every frame, the last two "
                                    "asynchronously gathered RGBA8 display images byte for byte; launches overlapped as in the timed region"}

    # what every rank saw, for the reader of a multi-GPU line: the slowest rank decides the frame time
    mine = {"rank": rank, "device": device, "rows": rows, "ms_per_step": own_elapsed / args.steps * 1e3,
            "ms_per_frame_synchronous": float(np.median(sync_ms[2:])),
            "ms_per_frame_single_in_flight": float(np.median(single_ms[2:])),
            "halo_wait_ms": (float(np.median(halo_wait)) if halo_wait else None),
            "pass_ms": {"gbuffer": float(np.median(gb_ms)), "to_rgba8": float(np.median(pbo_ms)), "primary": float(pass_ms[0]), "ris": float(pass_ms[1]),
                        "shadow_temporal": float(pass_ms[2]), "spatial_shade": float(pass_ms[3]),
                        **({"eaw_positions": eaw_level_ms[0], "eaw_levels": eaw_level_ms[1:]} if eaw_level_ms else {})},
            "rays_per_frame": local_rays / args.steps,
            "internal_streams": list(capi.internal_streams_info())}      # (priority level, calibration us of the chosen three, fastest candidate us) on this rank
    per_rank = [mine]
    set_phase("parity", mine=mine)                          # (the launcher quotes it if a later step fails)
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

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
        spatial_single_us = float(np.median(spatial_ms)) * 1e3
        spatial_us = spatial_b2b_us
        overlapped_us, overlapped_src = overlapped_kernel_us(args.config, "k_spatial_shade")
        alone_trace_us, alone_trace_src = trace_kernel_us(args.config, "k_spatial_shade", "kernel_stats.csv")
        in_frame_us = float(np.mean(spatial_in_frame_ms)) * 1e3 if spatial_in_frame_ms else None
        traffic, traffic_us, traffic_src = pmc_traffic(args.config, "k_spatial_shade")
        algo_bytes = ALGO_BYTES_PER_PIXEL * WIDTH * rows
        achieved = algo_bytes / (spatial_us * 1e-6) / 1e9
        out = {
            "metric": "Mrays/s (%dx%d ReSTIR-DI, 32 candidates, spatiotemporal reuse%s)" % (WIDTH, HEIGHT, " + 5-level EAW filter" if DENOISE else ""),
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_sustained": sustained["ms_per_step"] if sustained else None,
            "sustained": sustained,
            "ms_per_frame_synchronous": float(np.median(sync_ms[2:])),
            "ms_per_frame_single_in_flight": float(np.median(single_ms[2:])),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": cfg["workload"],
                       "baseline_config": args.config,
                       "camera": "orbit" if args.orbit else "static",
                       "sampler": "default engine (thrust minstd_rand, SAMPLER_USE_SOBOL false: the reference's default)" if sobol_num is None else
                                  "Sobol (src/sampler.h:9-36) over the build's own %d x 200 table" % sobol_num,
                       "launches": "asynchronous (rs_set_sync(0)): consecutive frames overlap on the library's auxiliary streams; GBuffer::render is walked "
                                   "together with the primary rays when the library measures that to be faster (it measures in untimed frames before the warm-up); "
                                   "ms_per_frame_synchronous is one frame alone with a synchronisation after every call, the reference's mode",
                       "tiling": (f"{world} row strips of " + "/".join(str(b - a) for a, b in bounds) + " rows (" +
                                  ("equal heights" if os.environ.get("BENCH_EVEN_STRIPS", "0") == "1" else "cost-balanced by measurement") + "), "
                                  "5 border rows of reservoirs + G-buffer id / normal / depth point-to-point (68 B/px)" +
                                  (", 32 G-buffer rows + 2 << level colour rows per EAW level" if DENOISE else "") + ", RGBA8 gather to rank 0") if world > 1 else "none",
                       "strip_driver": {"c": "librestir_hip rs_strips_frame / " + ("rs_strips_eaw_filter / " if DENOISE else "") + "rs_strips_gather_begin,_end (restir_amd/csrc/strips.hip)",
                                        "py": "restir_amd/tiling.py over torch.distributed (%s)" % backend_name,
                                        "none": "none (single GPU: rs_gbuffer_render + rs_restir_direct" + (" + rs_eaw_filter)" if DENOISE else ")")}[driver],
                       "transport": ("none" if driver == "none" else "torch.distributed " + backend_name if driver == "py" else
                                     "RCCL ncclSend / ncclRecv groups on the library stream, one group per frame (rs_comm_create_rccl_lib), ncclComm_t made by ncclCommInitRank from " + str(rccl.path)
                                     if transport == "rccl" else "host callbacks over torch.distributed gloo (rehearsal, not RCCL)"),
                       "rccl_ranks": (world if (driver == "c" and transport == "rccl") else (world if (driver == "py" and backend_name == "nccl" and world > 1) else 0)),
                       "strip_rows_per_rank": [b - a for a, b in bounds],
                       "strip_transfers_on": comm_choice,
                       "strip_transfers_trial_ms": comm_trial,
                       "strip_driver_fallback": fallback,
                       "halo_wait_ms_rank0": (float(np.median(halo_wait)) if halo_wait else None),
                       "rays_per_frame": total_rays / args.steps},
            "roofline": {"bound": "hbm", "kernel": "k_spatial_shade", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic if world == 1 else None,
                         "traffic_source": (("profiles/" + os.path.basename(traffic_src) + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; "
                                             "the kernel lasted %s us there)" % traffic_us) if traffic_src else None),
                         "algorithmic_bytes": algo_bytes, "kernel_us": spatial_us,
                         "kernel_us_how": "HIP events around 20 back-to-back launches of the pass on the library stream, / 20",
                         "kernel_us_single_launch": spatial_single_us,
                         # the same kernel inside the timed region's mode shares the CUs with the kernels of the other frames: its duration
                         # there comes from the committed kernel trace of this command with the frames overlapped (events would need the
                         # host to wait inside the frames, which drains the overlap they are meant to observe)
                         # three figures, each with its source.  (1) `frac` above: live, events around 20 warm back-to-back launches (the 190 MB
                         # working set fits the 256 MB Infinity Cache).  (2) the kernel alone in the committed rocprofv3 trace of this command:
                         "kernel_us_trace_alone": alone_trace_us if world == 1 else None,
                         "frac_trace_alone": (algo_bytes / (alone_trace_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if (alone_trace_us and world == 1) else None,
                         "trace_alone_source": ("profiles/" + os.path.basename(alone_trace_src) + " (rocprofv3 --kernel-trace, RS_SIDE_STREAM=0: every kernel alone on one stream; the "
                                                "measurement's own launches are k_spatial_shade_probe there)") if alone_trace_src else None,
                         # (3) inside the overlapped frames -- the mode ms_per_step times -- live: two events around the pass of each of 64 frames
                         # run after the timed region, read after the last of them (they include the events' own 2-4 us); and the same from the
                         # committed kernel trace of `bench.py --only-timed`, whose every frame is an overlapped one
                         "kernel_us_in_overlapped_frame": in_frame_us,
                         "frac_in_overlapped_frame": (algo_bytes / (in_frame_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if in_frame_us else None,
                         "in_overlapped_frame_how": "HIP events around the pass in each of 64 overlapped frames after the timed region (rs_restir_enable_timing 2), mean",
                         "kernel_us_in_overlapped_frame_trace": overlapped_us if world == 1 else None,
                         "frac_in_overlapped_frame_trace": (algo_bytes / (overlapped_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if (overlapped_us and world == 1) else None,
                         "overlapped_source": ("profiles/" + os.path.basename(overlapped_src)) if overlapped_src else None,
                         "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs if copy_gbs else None,
                         "note": "rank 0's strip; the timed span includes the wait for the halo rows" if world > 1 else "full frame"},
            "pass_ms": mine["pass_ms"],
        }
        if eaw_level_ms:
            eaw_us = sum(eaw_level_ms[1:]) * 1e3
            eaw_bytes = EAW_BYTES_PER_PIXEL_LEVEL * WIDTH * rows * 5
            out["roofline_eaw"] = {"bound": "hbm", "kernel": "k_wavelet_tiled (the five a-trous levels, steps 1..16)", "achieved": eaw_bytes / (eaw_us * 1e-6) / 1e9,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": eaw_bytes / (eaw_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                   "traffic": pmc_traffic(args.config, "k_wavelet_tiled")[0] if world == 1 else None,
                                   "algorithmic_bytes": eaw_bytes, "kernel_us": eaw_us, "level_us": [x * 1e3 for x in eaw_level_ms[1:]],
                                   "positions_us": eaw_level_ms[0] * 1e3,
                                   "note": "44 B per pixel and level (SURVEY.md 8d); `traffic` = mean HBM bytes of ONE level launch from the committed counters"}
        if world > 1:
            out["per_rank"] = per_rank
            exp = expected_compute_only(args.config, world)
            out["expected_compute_only_ms"] = exp["ms"] if exp else None
            out["expected_compute_only"] = exp
        fused, chains = timed_form                            # what the timed frames launched, as the library reported it then
        form = "one fused launch" if fused == 1 else "two launches"
        how = {-2: "not measured: a launch below three rounds of wave slots, or a form that is forced", -1: "measurement not finished within this run", 0: "measured", 1: "measured"}[backend.restir.launch_choice()]
        out["config"]["launch_choice"] = "%s (%s), chains on %d streams in turn" % (form, how, chains)
        out["config"]["calibration_frames_before_warmup"] = calibration_frames
        lvl, chosen_us, fastest_us = capi.internal_streams_info()
        out["config"]["internal_streams"] = {"priority_level": lvl, "calibration_us": chosen_us, "fastest_candidate_us": fastest_us,
                                             "note": "the library's three own streams, chosen by measurement next to the caller's stream (rs_internal_streams_info)"}
        if strips_parity is not None:
            out["parity" if world > 1 else "strips_parity"] = strips_parity       # N = 1 through the strip driver keeps the oracle's `parity` below
        if world == 1 and args.cpu_frames > 0 and sobol_num is None:
            n = args.cpu_frames
            out["cpu_baseline"], exact_images, exact_filtered = oracle_frames(cfg, sd, n, 1, True)
            got_images, got_filtered = gpu_frames(capi, cfg, sd, scene, n + 1)
            out["parity"] = compare_frames(cfg, exact_images, got_images, exact_filtered, got_filtered,
                                           "oracle/restir_oracle.c (libm mode: correctly rounded), frames looper 0.. of the benchmark workload from fresh "
                                           "reservoirs, radiance compared bit for bit" + ("; the filtered image within rtol 1e-5" if DENOISE else ""))
            del exact_images, exact_filtered
            _, glibc_images, glibc_filtered = oracle_frames(cfg, sd, n, 0, False)
            out["parity_glibc"] = compare_frames(cfg, glibc_images, got_images, glibc_filtered, got_filtered,
                                                 "oracle/restir_oracle.c in the mode the compiled reference pins (glibc cosf / sinf in the spatial taps, "
                                                 "src/restir.cu:47-56, src/mathUtil.h:128-132): stated tolerance mean per-pixel L1 < 1e-4 and at most 1e-3 of the "
                                                 "pixels beyond 1e-3")
            out["parity_glibc"]["within_tolerance"] = bool(out["parity_glibc"]["mean_l1"] < 1e-4 and out["parity_glibc"]["fraction_beyond_1e-3"] <= 1e-3)
            del glibc_images, glibc_filtered, got_images, got_filtered
            out["cpu_config1"] = cpu_config1(out["cpu_baseline"]["cores"])
            ref_loop = cpu_reference_loop(cfg, sd, out["cpu_baseline"]["cores"])
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
    set_phase("done", mine=mine)


if __name__ == "__main__":
    main()
