"""Soak of the lanes at the benchmark size: 300 views of the 1080p x 24-frame video per arrangement -- three lanes with second
streams in creation order, four single-stream lanes and two lanes with second streams placed by hardware queue -- injected noise,
every view compared with its sequential image (static image and masks bit for bit).  GPU box: python tools/soak_lanes.py"""
import sys
sys.path[:0]=['/root/repo','/root/repo/ml-pgdvs_amd']
import numpy as np, torch
from pgdvs_amd import synth, ops
from pgdvs_amd.instantiate import load_config
from pgdvs_amd.renderers.pgdvs_renderer import PGDVSRenderer
from pgdvs_amd.runtime import ResidentVideoRenderer
DEV="cuda:0"
T=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
H,W,S=1080,1920,24
v=synth.make_video(S,H,W,seed=1234)
cfg=load_config(static_renderer="geo", overrides={"engine.engine_cfg.render_cfg.dyn_pcl_remove_outlier": True, "engine.engine_cfg.render_cfg.st_render_pcl_pts_per_pixel": 3})
rc=cfg.engine.engine_cfg.render_cfg
model=PGDVSRenderer(cfg, render_cfg=rc).to(DEV).eval()
rvr=ResidentVideoRenderer(model, rc, T(v["rgbs"]), T(v["depths"]), T(v["dyn_masks"]), v["K3s"], v["c2ws"], lanes=3, side_streams=True)
datas=[synth.to_torch(synth.make_view(v,i,frac=0.4,seed=5),DEV) for i in (0,7,15,22)]
n0=rvr.calibrate(datas[0])
refs=[]
for d in datas:
    ret,_=rvr.render(d,0); rvr.join(); torch.cuda.synchronize()
    refs.append((ret["combined_rgb"].clone(), ret["geo_static_rgb"].clone(), ret["render_dyn_mask"].clone()))
for arrangement in ((3, True, False), (4, False, True), (2, True, True)):
    rvr.set_lanes(*arrangement)
    N=300
    bad=0
    ring=torch.empty((12,1,3,H,W),device=DEV)
    pending=[]
    for j in range(N):
        if len(pending)>=9:
            jj,ret=pending.pop(0)
            torch.cuda.synchronize()
            comb,st,dm=refs[jj%4]
            ok = int(ret["st_pcl_rgb_count"])==n0 and int(ret["geo_static_raster_status"])==0 and torch.equal(ret["geo_static_rgb"],st) and torch.equal(ret["render_dyn_mask"],dm) and torch.allclose(ring[jj%12],comb,rtol=0,atol=1e-5)
            bad += 0 if ok else 1
        pending.append((j, rvr.render(datas[j%4], j, out=ring[j%12])[0]))
    rvr.join(); torch.cuda.synchronize()
    for jj,ret in pending:
        comb,st,dm=refs[jj%4]
        ok = int(ret["st_pcl_rgb_count"])==n0 and torch.equal(ret["geo_static_rgb"],st) and torch.equal(ret["render_dyn_mask"],dm) and torch.allclose(ring[jj%12],comb,rtol=0,atol=1e-5)
        bad += 0 if ok else 1
    print("soak", arrangement, N, "views, mismatches:", bad)
