#!/usr/bin/env python3
"""tools/strip_period.py [CONFIG] -- the frame period of every rank of an N-way row split of BASELINE config 3, 4 or 5 on ONE GPU,
through the PRODUCT's strip driver: rs_strips_frame [+ rs_strips_eaw_filter] + GBuffer::update + tone map + rs_strips_gather_begin /
_end, exactly the frame() of `bench.py --gpus N`, over a stream-ordered transport that moves nothing (every host call and every
packing / unpacking launch of a real frame is made; RCCL's own group is not, and nothing travels: the image is wrong, only the TIMES
mean something).  Each rank of the split is timed alone on the card -- what that rank's GPU would spend per frame if the wire were free:
the compute-only bound of the strong-scaling curve.  Heights are cost-balanced by measurement (tiling.rebalance_bounds, as bench.py's
calibration does), N = 1 goes through the same driver (one-rank world).

    python tools/strip_period.py 3            # N = 1, 2, 4, 8; prints the table and, with OUT=path, writes it as JSON
  environment (A/B knobs, all optional):
    WORLDS=8                 which N to run (comma separated)
    ROWS=136,128,...         fixed strip heights for a single N (skips the balancing rounds)
    ROUNDS=3                 balancing rounds
    FRAMES=200               timed frames per rank
    TILE_SPLIT=n             rs_set_tile_split (negative: forced also for overlapped launches)
    STREAM_PLAN=c,s,m        rs_set_stream_plan(chain_streams, small_chains, shadow_on_main)
    RIS_TABLE_PIXELS=n       rs_set_ris_table_pixels
    COMM_STREAM=1            transfers on the driver's own stream
    PRE_STREAMS=n[d]         make n idle high-priority streams first (d: and destroy them): the order in which a process makes its
                             streams decides which of them run side by side (profiles/r05_ab_stream_levels_by_workload.log)
    RCCL=1                   create a one-rank ncclComm first (does RCCL's presence in the process move the period?)
    STREAM_LEVEL=-1|0|1|2    rs_set_internal_stream_priority (default: 1 for config 5, else 2 = automatic)
    STREAM_KIND=torch|torch_high|hip|null   the library stream: of torch's pool (default), high priority, a plain HIP stream, the legacy default stream
    GBUFFER_HALO=5|32        rs_strips_set_gbuffer_halo with the filter (default 32, as bench.py)
    DENOISE_STREAM=0|1       rs_set_denoise_stream: config 5's filter, tone map and display gather on the library stream (default) / on a stream of
                             their own, the chains on two streams
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from restir_amd import capi, scenes
from restir_amd.tiling import rebalance_bounds, strip_bounds

CONFIG = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W, H = (3840, 2160) if CONFIG == 4 else (1920, 1080)
DENOISE = CONFIG == 5
REUSE, TONEMAP = int(os.environ.get("REUSE", "3")), 2
NO_PBO = os.environ.get("NO_PBO", "0") == "1"            # sensitivity experiments: frames without the tone map and the display gather
WORLDS = [int(x) for x in os.environ.get("WORLDS", "1,2,4,8").split(",")]
ROUNDS = int(os.environ.get("ROUNDS", "3"))
FRAMES = int(os.environ.get("FRAMES", "200"))
FIXED = [int(x) for x in os.environ["ROWS"].split(",")] if os.environ.get("ROWS") else None

capi.init(0)
sd = scenes.bistro_class(seed=2, scale=1.0) if CONFIG == 5 else scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
KIND = os.environ.get("STREAM_KIND", "torch")       # what the library enqueues on: bench.py --gpus N hands it a stream of torch's pool
if KIND in ("torch", "torch_high"):
    stream = torch.cuda.Stream(priority=-1 if KIND == "torch_high" else 0)
    torch.cuda.set_stream(stream)
    capi.set_stream(stream.cuda_stream)
elif KIND == "hip":                                 # a stream made by hipStreamCreateWithFlags(hipStreamNonBlocking), as a C++ caller would
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    raw = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(raw), 1) == 0
    capi.set_stream(raw.value)
else:
    assert KIND == "null"                           # the legacy default stream
if os.environ.get("RCCL", "0") == "1":               # a one-rank ncclComm in the process, as bench.py --gpus N has one (RCCL makes streams of its own)
    from restir_amd.rccl import RcclComm
    _rccl = RcclComm(0, 1, lambda raw: raw)
if os.environ.get("PRE_STREAMS"):                   # experiment: make (and with a trailing "d" destroy) n high-priority streams first
    import ctypes
    _hip = ctypes.CDLL("libamdhip64.so")
    spec = os.environ["PRE_STREAMS"]
    made = []
    for _ in range(int(spec.rstrip("d"))):
        h = ctypes.c_void_p()
        assert _hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, -1) == 0
        made.append(h)
    if spec.endswith("d"):
        for h in made:
            assert _hip.hipStreamDestroy(h) == 0
capi.set_sync(False)
# what bench.py does: a frame with the denoiser on the library stream puts the library's own streams below it
capi.set_internal_stream_priority(int(os.environ.get("STREAM_LEVEL", "1" if DENOISE else "2")))
capi.set_denoise_stream(int(os.environ.get("DENOISE_STREAM", "0")) if DENOISE else 0)
if os.environ.get("TILE_SPLIT"):
    capi.set_tile_split(int(os.environ["TILE_SPLIT"]))
if os.environ.get("STREAM_PLAN"):
    capi.set_stream_plan(*[int(x) for x in os.environ["STREAM_PLAN"].split(",")])
if os.environ.get("RIS_TABLE_PIXELS"):
    capi.set_ris_table_pixels(int(os.environ["RIS_TABLE_PIXELS"]))
image = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda")
pbos = [torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
eaw = capi.EAWFilter(W, H, 5) if DENOISE else None
noop = lambda p, n, peer: None


def period(world, rank, bounds, frames=FRAMES):
    comm = capi.Comm(rank, world, noop, noop, None, None, stream_ordered=True)
    drv = capi.Strips(comm, W, H, [b[0] for b in bounds] + [H])
    if os.environ.get("COMM_STREAM", "0") == "1":
        drv.set_comm_stream(True)
    if DENOISE and world > 1 and os.environ.get("GBUFFER_HALO", "32") != "5":       # what bench.py does: the filter's G-buffer rows travel with the reservoir rows
        drv.set_gbuffer_halo(32)
    gbuf, restir = capi.GBuffer(W, H), capi.ReSTIR(W, H)
    y0, y1 = drv.y0, drv.y1
    st = {"n": 0}

    def frame():
        k = st["n"] % 2
        drv.frame(restir, scene, cam, gbuf, image.data_ptr(), 0, st["n"], REUSE)
        shown = image.data_ptr()
        if DENOISE:
            shown = drv.eaw_filter(eaw, gbuf, cam, image.data_ptr())
        gbuf.update(cam)
        st["n"] += 1
        if NO_PBO:
            return
        drv.gather_end(k)
        capi.copy_image_to_pbo(pbos[k].data_ptr() + y0 * W * 4, shown + y0 * W * 12, W, y1 - y0, TONEMAP, 1.0)
        drv.gather_begin(pbos[k].data_ptr(), 4, 0, k)

    for _ in range(36):                       # (N = 1: past the library's measured launch choice, frames 2-34 ...
        frame()
    capi.synchronize(); torch.cuda.synchronize()
    for _ in range(2):                        # ... which is read at the first frame end after the last stamp has been reached)
        frame()
    capi.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        frame()
    t1 = time.perf_counter()
    capi.synchronize(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    drv.gather_end(0); drv.gather_end(1)
    capi.synchronize()
    form = restir.last_launch()
    drv.destroy(); comm.destroy(); restir.destroy(); gbuf.destroy()
    return (t2 - t0) / frames * 1e3, (t1 - t0) / frames * 1e3, form


table = {"config": CONFIG, "width": W, "height": H, "frames": FRAMES, "worlds": {},
         "what": "frame period of each rank alone on one MI355X through rs_strips_frame over a transport that moves nothing (compute only)"}
min_rows = 32 if DENOISE else 8
t1 = None
for n in WORLDS:
    bounds = [strip_bounds(H, n, r) for r in range(n)]
    if FIXED and len(FIXED) == n:
        ys = [0]
        for r in FIXED:
            ys.append(ys[-1] + r)
        assert ys[-1] == H, "ROWS must sum to the frame height"
        bounds = [(ys[i], ys[i + 1]) for i in range(n)]
    rounds = 0 if (n == 1 or (FIXED and len(FIXED) == n)) else ROUNDS
    best = None
    for it in range(rounds + 1):
        res = [period(n, r, bounds) for r in range(n)]
        ms = [x[0] for x in res]
        host = max(x[1] for x in res)
        print("config %d N=%d %s: rows %s  ms %s  max %.4f  host enqueue %.3f  launch (fused, chains) %s" % (
            CONFIG, n, "even" if it == 0 and not FIXED else "balanced %d" % it, " ".join(str(b - a) for a, b in bounds),
            " ".join("%.4f" % t for t in ms), max(ms), host, res[0][2]), flush=True)
        if it == 0:
            print("   internal streams (level, calibration us, fastest candidate us):", capi.internal_streams_info(), flush=True)
        if best is None or max(ms) < best["max_ms"]:
            best = {"rows": [b - a for a, b in bounds], "ms": ms, "max_ms": max(ms), "host_enqueue_ms": host}
        if it < rounds:
            bounds = rebalance_bounds(bounds, ms, H, min_rows=min_rows)
    if n == 1:
        t1 = best["max_ms"]
    if t1:
        best["speedup_compute_only"] = t1 / best["max_ms"]
        print("config %d N=%d: max over ranks %.4f ms -> %.2fx compute-only (N=1 %.4f ms)" % (CONFIG, n, best["max_ms"], t1 / best["max_ms"], t1), flush=True)
    table["worlds"][str(n)] = best
if os.environ.get("OUT"):
    with open(os.environ["OUT"], "w") as fh:
        json.dump(table, fh, indent=1)
