"""multigpu._spatial_run (the timed loop of bench.py --gpus N) under a one-rank RCCL group on one card: a smoke run of the
code the driver's multi-GPU bench goes through (statistics, the line's objects), not a measurement."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import torch.distributed as dist
import bench
from subzero_amd import multigpu
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
S, P = bench.build_workload(4000)
el, st = multigpu._spatial_run(S, P, 0, 0, 1, 20, 5, torch, dist)
print(json.dumps({"ms_per_step": el / 20 * 1e3, **{k: v for k, v in st.items() if k != "strip_edges_m"}}))
dist.destroy_process_group()
