"""Decode-step latency after a cfg2 prefill (greedy, batch 1): eager per-op path vs the captured hipGraph (quickvideo_amd/decode.py),
with the weight-stream bound beside it."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from quickvideo_amd.decode import GraphDecoder
dev = torch.device("cuda:0")
spec, cfg, plan, eng, embeds, pos, T = bench.build_workload(os.environ.get("QP_CFG", "cfg2"), dev, 0, 1)
tok = bench.run_step(eng, plan, embeds, pos)
torch.cuda.synchronize()
len0, pos0 = list(eng.arena.len), eng.seq_pos
emb = eng.embed_tokens(tok.view(1))
for steps in (8, 64):
    eng.arena.len, eng.seq_pos = list(len0), pos0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        logits = eng.decode_step(emb, rope_delta=0)
        nxt = torch.argmax(logits)
        emb = eng.embed_tokens(nxt.view(1))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"eager  {steps} decode steps: {dt / steps * 1e3:.2f} ms/token ({steps / dt:.1f} tok/s), KV {eng.arena.len[0]} rows/layer", flush=True)
dec = GraphDecoder(eng)
wbytes = sum(t.numel() * 2 for lw in eng.w.layers for t in (lw.w_qkv, lw.w_o, lw.w_gate_up, lw.w_down)) + eng.w.lm_head.numel() * 2
kvbytes = sum(2 * eng.hkv * n * eng.D * 2 for n in len0)
for steps in (8, 64, 256):
    eng.arena.len, eng.seq_pos = list(len0), pos0
    dec.generate(int(tok), 2, 0)                                   # capture / warm
    eng.arena.len, eng.seq_pos = list(len0), pos0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    toks = dec.generate(int(tok), steps, 0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = len(toks)                                              # bounded by the room left in the KV arena
    ms = dt / steps * 1e3
    print(f"graph  {steps} decode steps: {ms:.3f} ms/token ({steps / dt:.1f} tok/s); weights {wbytes / 1e9:.2f} GB + KV {kvbytes / 1e9:.3f} GB per token "
          f"-> {(wbytes + kvbytes) / ms / 1e9:.2f} TB/s", flush=True)
