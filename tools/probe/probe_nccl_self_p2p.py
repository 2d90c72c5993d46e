"""Can a 1-GPU box execute RCCL point-to-point at all?  (VERDICT r3 Missing #2: the layer pipeline's send/recv has had zero RCCL contact;
a 1-rank group cannot talk to a peer, but NCCL allows a rank to send to ITSELF inside one group call.)  Tries, on a 1-rank nccl group:
batch_isend_irecv([isend(t, 0), irecv(r, 0)]) on device tensors of the pipeline's hand-off shape, and plain isend/irecv."""
import os, socket, sys, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1])); s.close()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.randn(2240, 3584, device=dev).to(torch.bfloat16); r = torch.zeros_like(t)
for name, fn in (("batch_isend_irecv", lambda: [w.wait() for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, 0), dist.P2POp(dist.irecv, r, 0)])]),
                 ("isend+irecv", lambda: [w.wait() for w in (dist.irecv(r, 0), dist.isend(t, 0))])):
    r.zero_()
    try:
        fn(); torch.cuda.synchronize()
        print(name, "OK" if torch.equal(t, r) else "WRONG DATA", flush=True)
    except Exception as e:
        print(name, "FAILED:", type(e).__name__, str(e)[:200], flush=True)
dist.destroy_process_group()
