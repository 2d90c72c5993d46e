import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
spec, cfg, plan, eng, embeds, pos, T = bench.build_workload("cfg2", dev, 0, 1)
for _ in range(2): bench.run_step(eng, plan, embeds, pos)
torch.cuda.synchronize()
for rep in range(3):
    eng.reset(); start = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for n in plan.tokens:
        eng.prefill_group(embeds[start:start + n], pos[:, start:start + n]); start += n
    torch.cuda.synchronize(); t1 = time.perf_counter()
    logits = eng.prefill_tail(embeds[start:], pos[:, start:])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    # host-only time of the tail (enqueue without waiting)
    print(f"groups {1e3*(t1-t0):.1f} ms, tail {1e3*(t2-t1):.2f} ms")
