"""Will RCCL run TWO ranks on ONE GPU?  (NCCL refuses "duplicate GPU"; if RCCL accepts it, a 1-GPU box can exercise the real N = 2 path:
collectives, point-to-point, the pipeline's front-end group.)  Spawns 2 processes on cuda:0 with backend nccl."""
import os, socket, sys, torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", 0))
        t = torch.full((1024,), float(rank + 1), device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(rank, "all_reduce ->", float(t[0]), flush=True)
        x = torch.full((2240, 3584), float(rank), device="cuda:0", dtype=torch.bfloat16)
        if rank == 0:
            dist.send(x, dst=1)
        else:
            dist.recv(x, src=0)
        torch.cuda.synchronize()
        print(rank, "p2p ok", float(x[0, 0]), flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print(rank, "FAILED:", type(e).__name__, str(e)[:300], flush=True)


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(port,), nprocs=2, join=True)
