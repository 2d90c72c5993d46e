"""How long does the prompt tail (a few dozen tokens over the pruned cache) take after a cfg2 prefill?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
spec, cfg, plan, eng, embeds, pos, T = bench.build_workload(os.environ.get("QP_CFG", "cfg2"), dev, 0, 1)
for it in range(3):
    eng.reset(); start = 0
    for n in plan.tokens:
        eng.prefill_group(embeds[start:start + n], pos[:, start:start + n]); start += n
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record(); logits = eng.prefill_tail(embeds[start:], pos[:, start:]); e.record(); torch.cuda.synchronize()
    print(f"tail of {plan.tail_len} tokens over {eng.arena.len[0]} cached rows: {s.elapsed_time(e):.3f} ms", flush=True)
