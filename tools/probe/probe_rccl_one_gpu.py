"""Can two RCCL ranks share the one GPU of a test box?  No: `ncclInvalidUsage: Duplicate GPU detected : rank 1 and rank 0 both on CUDA
device a4000` (NCCL 2.26.6 in torch 2.10+rocm7.0, MI355X).  Hence the multi-rank GPU runs of this repo use gloo with
QP_BENCH_SINGLE_DEVICE=1 (host-staged hand-offs) and RCCL itself is first exercised on a multi-GPU node.
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/probe/probe_rccl_one_gpu.py"""
import os, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
x = torch.ones(1024, device="cuda:0") * (rank + 1)
dist.all_reduce(x)
torch.cuda.synchronize()
print("rank", rank, "allreduce ok", x[0].item(), flush=True)
if rank == 0:
    dist.send(x, dst=1)
else:
    y = torch.empty_like(x); dist.recv(y, src=0); torch.cuda.synchronize(); print("recv ok", y[0].item(), flush=True)
dist.destroy_process_group()
