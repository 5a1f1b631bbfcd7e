import os, sys, time
if "--stats" in sys.argv: os.environ["PGDVS_KNN_STATS"]="1"
import pathlib; R=pathlib.Path(__file__).resolve().parent.parent; sys.path.insert(0,str(R)); sys.path.insert(0,str(R/"ml-pgdvs_amd"))
import numpy as np, torch, ctypes as C
from pgdvs_amd import ops, synth, _lib
from pgdvs_amd.instantiate import AttrDict
from pgdvs_amd.renderers.pgdvs_renderer_dyn import PGDVSDynamicRenderer
dev="cuda:0"
S,H,W=3,1080,1920
v=synth.make_video(S,H,W,seed=1234)
d=synth.to_torch(synth.make_view(v,0,seed=5),dev)
cams=ops.cam_prep(d["flat_cam_src_temporal"]); camt=ops.cam_prep(d["flat_cam_tgt"])
times=torch.cat([d["time_src_temporal"][:,:2],d["time_tgt"][:,:1]],1).contiguous()
mask_eff, valid, pcl, rgbf = ops.dyn_warp(d["dyn_mask_src_temporal"][0,0,...,0], d["flow_fwd_occ_mask"][0,...,0], False, d["flow_fwd"][0], d["depth_src_temporal"][0,0,...,0], d["depth_src_temporal"][0,1,...,0], d["rgb_src_temporal"][0,0], d["rgb_src_temporal"][0,1], cams[0,0], cams[0,1], times[0])
idx,cnt=ops.compact_u8(valid); pts=ops.gather_rows(pcl.reshape(-1,3), idx, cnt)
n=int(cnt.item()); print("n",n)
lib=_lib.load()
ws=torch.empty(lib.pgdvs_knn_workspace_bytes(pts.shape[0]),dtype=torch.uint8,device=dev)
out=torch.empty(pts.shape[0],device=dev)
def run():
    lib.pgdvs_knn_mean_dist(C.c_void_p(pts.data_ptr()),C.c_void_p(cnt.data_ptr()),pts.shape[0],50,C.c_void_p(out.data_ptr()),2,C.c_void_p(ws.data_ptr()),ws.numel(),C.c_void_p(0))
run(); torch.cuda.synchronize()
t=time.perf_counter(); run(); torch.cuda.synchronize(); print("ms",(time.perf_counter()-t)*1e3)
lib.pgdvs_prof_enable(1); run(); run(); torch.cuda.synchronize(); lib.pgdvs_prof_enable(0)
buf=C.create_string_buffer(1<<14); lib.pgdvs_prof_report(buf,len(buf)); print(buf.value.decode())
st=None
gp=ws[256:256+48].cpu().numpy(); print("mn,h,inv_h", gp[:20].view(np.float32), "G,ncells,n", gp[20:40].view(np.int32))
p=pts[:n].cpu().numpy(); print("bbox", p.min(0), p.max(0))
