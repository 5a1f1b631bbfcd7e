import sys, time, pathlib
R0 = pathlib.Path("/root/repo"); sys.path.insert(0, str(R0)); sys.path.insert(0, str(R0 / "ml-pgdvs_amd"))
import torch
from pgdvs_amd import ops, _lib
from pgdvs_amd.models.gnt.models.transformer_network import GNT
dev="cuda:0"; torch.manual_seed(0)
net = GNT(netwidth=64, transformer_depth=8).to(dev).eval()
R,S=1024,256
q = torch.randn(R,S,64,device=dev)
layer = net.view_selftrans[0]
def run():
    with torch.no_grad():
        return ops.gnt_ray_layer(layer, q, True)
outs={}
for mode in (1,0,1,0):
    ops.set_option("gnt_ray_fp32", mode)
    o=run(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): o=run()
    e1.record(); torch.cuda.synchronize()
    print("gnt_ray_fp32 =",mode, "ms per ray layer (attn + FF):", e0.elapsed_time(e1)/20)
    outs[mode]=o
a,b=outs[1],outs[0]
if isinstance(a,(tuple,list)):
    for x,y in zip(a,b):
        if torch.is_tensor(x): print("max abs diff", float((x-y).abs().max()), "max abs", float(x.abs().max()))
else:
    print("max abs diff", float((a-b).abs().max()))
