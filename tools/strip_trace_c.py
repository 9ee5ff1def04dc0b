#!/usr/bin/env python3
"""tools/strip_trace_c.py [WORLD RANK] -- one strip of an N-way split through the C-ABI strip driver (rs_strips_frame, tone map,
rs_strips_gather_begin / _end) with a stream-ordered no-op transport, 80 frames, to be run under the profiler with the interpreter
after `--`:   rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/strip_trace_c.py 8 3
tools/strip_trace_report.py condenses the trace (per queue the busy time per frame, per kernel the mean duration)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from restir_amd import capi, scenes

CONFIG = int(os.environ.get("CONFIG", "3"))           # BASELINE config 3 (default), 4 (3840x2160) or 5 (Bistro-class + EAW filter)
W, H = (3840, 2160) if CONFIG == 4 else (1920, 1080)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
capi.init(0)
sd = scenes.bistro_class(seed=2, scale=1.0) if CONFIG == 5 else scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
capi.set_sync(False)
if CONFIG == 5:
    capi.set_internal_stream_priority(1)
eaw = capi.EAWFilter(W, H, 5) if CONFIG == 5 else None
comm = capi.Comm(rank, world, lambda p, n, peer: None, lambda p, n, peer: None, None, None, stream_ordered=True)
bounds = None
if os.environ.get("ROWS"):                                # strip heights of all ranks (default: equal heights)
    ys = [0]
    for r in os.environ["ROWS"].split(","):
        ys.append(ys[-1] + int(r))
    assert len(ys) == world + 1 and ys[-1] == H
    bounds = ys
drv = capi.Strips(comm, W, H, bounds)
if eaw and world > 1 and os.environ.get("GBUFFER_HALO", "32") != "5":
    drv.set_gbuffer_halo(32)
gbuf, restir = capi.GBuffer(W, H), capi.ReSTIR(W, H)
image = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda")
pbos = [torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
y0, y1 = drv.y0, drv.y1
def frame(n):
    k = n % 2
    drv.frame(restir, scene, cam, gbuf, image.data_ptr(), 0, n, 3)
    shown = drv.eaw_filter(eaw, gbuf, cam, image.data_ptr()) if eaw else image.data_ptr()
    gbuf.update(cam)
    drv.gather_end(k)
    capi.copy_image_to_pbo(pbos[k].data_ptr() + y0 * W * 4, shown + y0 * W * 12, W, y1 - y0, 2, 1.0)
    drv.gather_begin(pbos[k].data_ptr(), 4, 0, k)


for n in range(36):                                      # (past the library's measured launch choice, frames 2-34, and the frame end that reads it)
    frame(n)
capi.synchronize(); torch.cuda.synchronize()
for n in range(36, 40):
    frame(n)
capi.synchronize(); torch.cuda.synchronize()
# Under the profiler the HOST needs longer per frame than a 1/8 strip's GPU work (every dispatch is intercepted), and a host-bound
# timeline says nothing about the GPU's own pace: SPIN_MS (default 60) of a spinning kernel on the library stream hold the frames'
# temporal / spatial passes back while the host enqueues all 60 timed frames; what runs after the spin is paced by the GPU alone
# (the first chains_in_flight + 1 chains do not wait for the library stream and run during the spin).
spin_ms = float(os.environ.get("SPIN_MS", "60"))
if spin_ms > 0:
    torch.cuda._sleep(int(spin_ms * 1e-3 * 2.4e9))         # shader-clock cycles (2.5 ms per 6e6 measured)
for n in range(40, 100):
    frame(n)
capi.synchronize(); torch.cuda.synchronize()
