"""Decode-step latency after a cfg2 prefill (greedy, batch 1): how launch-bound is the single-token path?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
spec, cfg, plan, eng, embeds, pos, T = bench.build_workload("cfg2", dev, 0, 1)
tok = bench.run_step(eng, plan, embeds, pos)
torch.cuda.synchronize()
emb = eng.embed_tokens(tok.view(1))
for steps in (8, 64):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        logits = eng.decode_step(emb, rope_delta=0)
        nxt = torch.argmax(logits)
        emb = eng.embed_tokens(nxt.view(1))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{steps} decode steps: {dt / steps * 1e3:.2f} ms/token ({steps / dt:.1f} tok/s), KV {eng.arena.len[0]} tokens/layer")
