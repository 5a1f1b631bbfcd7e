#!/usr/bin/env python3
"""RCCL sanity of the per-step image gather with a single rank (all a 1-GPU box can run): the
gather into slices of the preallocated stack, asynchronously, from a side stream."""
import os, pathlib, sys
R0 = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R0 / "ml-pgdvs_amd"))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from pgdvs_amd import dist as pdist
g = pdist.AsyncImageGather(dst=0, n_steps=3, like=torch.empty(1, 3, 1080, 1920, device=dev))
g.on = True  # a single rank normally skips the collective
g.stack = torch.empty((3, 1, 1, 3, 1080, 1920), device=dev)
side = torch.cuda.Stream()
imgs = []
for j in range(3):
    with torch.cuda.stream(side):
        img = torch.full((1, 3, 1080, 1920), float(j), device=dev)
        imgs.append(img)
        g.submit(img)
torch.cuda.current_stream().wait_stream(side)
out = g.finish()
torch.cuda.synchronize()
assert out.shape == (3, 3, 1080, 1920) and [float(out[j, 0, 0, 0]) for j in range(3)] == [0.0, 1.0, 2.0], out.shape
assert out.data_ptr() == g.stack.data_ptr()
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print("NCCL_GATHER_OK")
