"""bench.py's control-plane calls (init_process_group("nccl"), barrier, max-over-ranks all-reduce on a device tensor)
with one rank on one GPU: the RCCL path cannot be rehearsed with two ranks on a one-GPU box (RCCL refuses two ranks
on one device), so at least its single-rank form is run on a real MI355X.   python tools/check_rccl_control_plane.py"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "anemoi-rust_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29599")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
from anemoi_amd.shard import max_over_ranks
print("max_over_ranks over RCCL:", max_over_ranks(1.25, dist, torch.device("cuda", 0)))
dist.barrier(); dist.destroy_process_group(); print("rccl ok")
