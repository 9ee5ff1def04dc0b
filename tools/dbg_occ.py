import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from restir_amd import capi as hip
from oracle import binding as ob
from tests.common import get_scene
from tests.test_gpu_parity import _shadow_like_segments
hip.init(0)
sd = get_scene("sponza:0.03")
base = hip.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials)
t = base.host_desc()
rng = np.random.default_rng(5)
boxes = t["boxes"].copy()
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
if mode in ("grow", "both"):
    pick = rng.choice(len(boxes), len(boxes) // 3, replace=False)
    boxes[pick, :3] -= rng.uniform(0, 0.2, (len(pick), 3)).astype(np.float32)
    boxes[pick, 3:] += rng.uniform(0, 0.2, (len(pick), 3)).astype(np.float32)
if mode in ("shrink", "both"):
    shrink = rng.choice(len(boxes), len(boxes) // 10, replace=False)
    mid = 0.5 * (boxes[shrink, :3] + boxes[shrink, 3:])
    boxes[shrink, :3] = 0.5 * (boxes[shrink, :3] + mid); boxes[shrink, 3:] = 0.5 * (boxes[shrink, 3:] + mid)
t["boxes"] = boxes
args = (sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials, t)
fast = hip.Scene.from_tables(*args)
os.environ["RS_NO_OCCLUSION_TREE"] = "1"
slow = hip.Scene.from_tables(*args)
del os.environ["RS_NO_OCCLUSION_TREE"]
segh = _shadow_like_segments(sd, 200000, 12)
seg = torch.from_numpy(segh).cuda()
a = hip.trace_occlusion(fast, seg).cpu().numpy()
b = hip.trace_occlusion(slow, seg).cpu().numpy()
bad = np.nonzero(a != b)[0]
print("mode", mode, "mismatches", len(bad), "of", len(a), "first", bad[:20], "k =", len(a) // 8)
osc = ob.Scene(sd.vertices, sd.normals, sd.texcoords, sd.material_ids, sd.materials,
               prebuilt=(t["light_prim_ids"], t["light_radiance"], np.zeros(len(t["light_prim_ids"]), np.float32), t["light_prob"], t["light_fail"], t["sum_power"], boxes, t["nodes"]))
idx = np.concatenate([bad[:2000], np.arange(0, len(a), 50)])
c = osc.test_occlusion(segh[idx])
print("fast != oracle:", int((a[idx] != c).sum()), " slow != oracle:", int((b[idx] != c).sum()), "of", len(idx))
for i in bad[:8]:
    print(i, "fast", a[i], "slow", b[i], "oracle", osc.test_occlusion(segh[i:i+1])[0], segh[i])
