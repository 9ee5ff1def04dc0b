#!/usr/bin/env python3
"""tools/strip_trace.py RANK -- runs strip RANK of the 8-way split alone for 60 frames, to be run under the profiler with the
interpreter itself after `--` (never this script directly: its `#!/usr/bin/env` line would be an exec hop under a preloaded profiler):

    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python $REPO/tools/strip_trace.py RANK

tools/strip_trace_report.py condenses the trace: per queue the busy time per frame, per kernel the mean duration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from restir_amd import capi, scenes
from restir_amd.tiling import HipBackend, StripRenderer

W, H = 1920, 1080
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 3
capi.init(0)
sd = scenes.sponza_class(seed=1, scale=1.0)
scene = capi.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
cam = capi.camera_update(sd.camera(W, H))
backend = HipBackend(capi, scene, cam, W, H)
capi.set_sync(False)
pbo = torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda")
ROWS8 = [152, 128, 120, 80, 88, 104, 176, 232]
b, y = [], 0
for r in ROWS8:
    b.append((y, y + r)); y += r
s = StripRenderer(backend, 8, rank, H, bounds=b)
s.start_halo_exchange = lambda: ([], [], [])
for _ in range(60):
    s.frame(3, 0)
    capi.copy_image_to_pbo(pbo.data_ptr(), backend.image.data_ptr() + s.y0 * W * 12, W, s.y1 - s.y0, 2, 1.0)
torch.cuda.synchronize()
