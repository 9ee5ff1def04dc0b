import sys; sys.path.insert(0,'.')
import torch, numpy as np, ctypes as C
from restir_amd import capi, scenes
from tests.common import HipRenderer, get_scene
capi.init(0)
sd=get_scene('sponza:1.0')
h=HipRenderer(capi,sd,1920,1080)
for f in range(3):
    h.frame(3)
# pass times
h.restir.enable_timing(True)
capi.set_sync(False)
for f in range(3):
    h.gbuf.render(h.scene,h.cam); h.restir.direct(h.scene,h.cam,h.gbuf,h.image.data_ptr(),0,10+f,3); torch.cuda.synchronize(); print(h.restir.pass_times())
