#!/usr/bin/env python
"""bench.py — prefill tokens/s (+ TTFT) of the QuickPrefill hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4s|tiny]
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" = one full pass of the hot path over one synthetic video: every group through all decoder layers
(QKV + M-RoPE/append + MFMA attention over the pruned prefix + key-norm select + KV gather + MLP) plus the
prompt tail up to the first-token logits.  Inputs (ViT-output embeddings, position ids, weights) are resident in
HBM when the timed region starts.  Default workload = BASELINE.json configs[1]:
Qwen2-VL-7B, 64 frames (560x1008), group_size 16 -> 4 groups x 5760 tokens, key-norm rho=0.5, 1 x MI355X.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant hand-written kernel =
the MFMA prefill attention, measured live with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle
timed on the host cores over a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from quickvideo_amd import planner  # noqa: E402
from quickvideo_amd.engine import QuickPrefillEngine, sp_row_ranges  # noqa: E402
from quickvideo_amd.lvu_config import LVUConfig, effective_k  # noqa: E402
from quickvideo_amd.spec import PRESETS  # noqa: E402
from quickvideo_amd.weights import DecoderWeights, pp_layer_split  # noqa: E402

CONFIGS = {
    # name: (model, frames, frame_h, frame_w, group_size, rho, prefix, tail)
    "cfg1": ("qwen2-vl-2b", 16, 560, 1008, 16, 1.0, 15, 30),      # BASELINE.json configs[0]: 2B, one group, no pruning
    "cfg2": ("qwen2-vl-7b", 64, 560, 1008, 16, 0.5, 15, 30),
    "cfg3": ("qwen2-vl-7b", 256, 280, 504, 32, 0.25, 15, 30),
    "cfg4s": ("qwen2-vl-7b", 720, 392, 560, 16, 0.5, 15, 30),     # 1/10 of the 1-hour video (100k tokens)
    "cfg4": ("qwen2-vl-7b", 7200, 392, 560, 16, 0.5, 15, 30),      # synthetic 1-hour video, ~1M vision tokens
    "cfg5": ("qwen2-vl-72b", 512, 224, 420, 16, 0.5, 15, 30),      # 72B: TP=8 in BASELINE.json; also fits ONE MI355X (145 GB of 288 GB)
    "tiny": ("tiny", 16, 112, 168, 4, 0.5, 5, 7),
}
PEAK_BF16_TFLOPS = 2500.0     # dense MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


class TimedOps:
    """Proxy over QuickPrefillOps that brackets chosen operators with HIP events on the launch stream."""

    def __init__(self, ops, names):
        self._ops, self._names, self.events = ops, set(names), {n: [] for n in names}

    def __getattr__(self, name):
        fn = getattr(self._ops, name)
        if name not in self._names:
            return fn

        def timed(*a, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **kw)
            e.record()
            self.events[name].append((s, e))
            return r
        return timed

    def totals_ms(self):
        torch.cuda.synchronize()
        return {n: (sum(s.elapsed_time(e) for s, e in ev), len(ev)) for n, ev in self.events.items()}


def choose_layout(n_groups, world):
    """(pp, sp) with pp * sp == world for `--parallel auto`: a layer pipeline of pp stages, each a group-token parallel group of
    sp ranks.  Model: the pipe is busy G / (G + pp - 1) of the time; an sp group of s ranks runs at EFF_SP[s] of a single GPU
    (GEMMs at M = n/s, exchange, replicated prune; measured / estimated at cfg2 sizes)."""
    EFF_SP = {1: 1.0, 2: 0.92, 4: 0.75, 8: 0.5}
    best, best_eff = (1, world), -1.0
    pp = 1
    while pp <= world:
        sp = world // pp
        if pp * sp == world and sp in EFF_SP:
            eff = n_groups / (n_groups + pp - 1) * EFF_SP[sp]
            if eff > best_eff + 1e-9:
                best, best_eff = (pp, sp), eff
        pp *= 2
    return best


def build_workload(name, device, rank, world, seed=0, parallel="sp", layout=None):
    model, frames, fh, fw, gs, rho, prefix, tail = CONFIGS[name]
    spec = PRESETS[model]
    gh, gw = fh // 14, fw // 14
    n_video = (frames // 2) * (gh // 2) * (gw // 2)
    T = prefix + n_video + tail
    plan = planner.plan_groups(frames, gs, gh, gw, prefix, T)
    pos, delta = planner.mrope_positions(prefix, (frames // 2, gh, gw), tail, temporal_scale=spec.temporal_scale)
    cfg = LVUConfig(model, top_p=rho, video_group_size=gs, num_frames=frames)
    tp = parallel == "tp" and world > 1
    pp_n, sp_n = layout if layout is not None else (1, 1)
    stage, sp_rank = rank // sp_n, rank % sp_n                                   # ranks of a stage are consecutive
    weights = DecoderWeights.synthetic(spec, device, seed=seed, tp_rank=rank if tp else 0, tp_size=world if tp else 1,
                                       layer_range=pp_layer_split(spec.n_layers, pp_n, stage) if pp_n > 1 else None)
    kept = sum(effective_k(n, cfg, 0, spec.n_layers) or n for n in plan.tokens)
    cap = kept + plan.tail_len + 64
    eng = QuickPrefillEngine(weights, cfg, capacity=cap, max_group_tokens=max(plan.tokens + [plan.tail_len]), device=device,
                             sp_rank=sp_rank, sp_size=sp_n, pp_rank=stage, pp_size=pp_n,
                             pp_peers=[s * sp_n + sp_rank for s in range(pp_n)] if pp_n > 1 else None)
    eng.rope_delta = int(delta)                                  # decode positions continue at sequence index + delta
    g = torch.Generator(device=device); g.manual_seed(1234)      # same embeddings on every TP rank
    # synthetic ViT output / text embeddings: N(0, 1) scaled like embedding rows (the ViT front end is bench'd separately)
    embeds = (torch.randn(T, spec.hidden, generator=g, device=device, dtype=torch.float32) * 0.5).to(torch.bfloat16)
    pos_d = torch.from_numpy(pos).to(device)
    return spec, cfg, plan, eng, embeds, pos_d, T


def run_step(eng, plan, embeds, pos):
    eng.reset()
    start = 0
    for n in plan.tokens:
        eng.prefill_group(embeds[start:start + n], pos[:, start:start + n])
        start += n
    logits = eng.prefill_tail(embeds[start:], pos[:, start:])
    tok = torch.argmax(logits) if logits is not None else torch.zeros((), dtype=torch.int64, device=embeds.device)
    if eng.pp_size > 1:                  # layer pipeline: the last stage holds the logits; the token returns to every stage
        torch.distributed.broadcast(tok, src=torch.distributed.get_world_size() - 1)
    return tok                           # first generated token id (stays on device; .item() would be the TTFT point)


def decode_leg(eng, first_token: int, n_tokens: int = 32):
    """Greedy decode after the prefill (a10): ms per token of the captured hipGraph step (quickvideo_amd/decode.py) and the HBM
    stream it is bounded by (every decoder weight + lm_head once, every cached K/V row once per token)."""
    from quickvideo_amd.decode import GraphDecoder
    if not GraphDecoder.supported(eng):
        return None
    dec = GraphDecoder(eng)
    len0, pos0 = list(eng.arena.len), eng.seq_pos
    n_tokens = min(n_tokens, eng.arena.capacity - max(len0))
    if n_tokens < 1:
        return None
    dec.generate(first_token, min(4, n_tokens), eng.rope_delta)          # capture + warm
    eng.arena.len, eng.seq_pos = list(len0), pos0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks = dec.generate(first_token, n_tokens, eng.rope_delta)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / len(toks) * 1e3
    eng.arena.len, eng.seq_pos = list(len0), pos0
    wbytes = sum(t.numel() * 2 for lw in eng.w.layers for t in (lw.w_qkv, lw.w_o, lw.w_gate_up, lw.w_down)) + eng.w.lm_head.numel() * 2
    kvbytes = sum(2 * eng.hkv * n * eng.D * 2 for n in len0)
    tbs = (wbytes + kvbytes) / (ms * 1e-3) / 1e12
    return {"ms_per_token": round(ms, 3), "tokens": len(toks), "mode": "hipGraph replay per token", "hbm_bytes_per_token": wbytes + kvbytes,
            "achieved_tb_s": round(tbs, 2), "frac_of_hbm_peak": round(tbs / 8.0, 3), "kv_rows_per_layer": len0[0]}


def flops_and_bytes(spec, cfg, plan, tp_size):
    """Algorithmic FLOPs of one step (F_lin per token, F_att(g) = 4 L Hq D (n P + n(n+1)/2)) and the
    prune-path bytes in the unfused convention (B_prune = n Hkv D 2 + 2 (k Hkv D 2 2) + 4k per layer per group)."""
    L = spec.n_layers
    lin = att = prune_bytes = 0.0
    P = 0
    for n in plan.tokens:
        lin += spec.linear_flops_per_token() * n
        att += spec.attn_flops(n, P)
        k = effective_k(n, cfg, 0, L)
        if k is not None:
            prune_bytes += L * (n * spec.kv_dim * 2 + 2 * (k * spec.kv_dim * 2 * 2) + 4 * k)
        P += k if k is not None else n
    lin += spec.linear_flops_per_token() * plan.tail_len          # prompt tail: no pruning
    att += spec.attn_flops(plan.tail_len, P)
    return lin, att, prune_bytes


def cpu_baseline(name, sample_layers=2, sample_tokens=2880):
    """Time the CPU oracle (oracle/qp_oracle.py — the checker, used here only as the reported baseline) on a
    bounded sample: `sample_layers` decoder layers of the SECOND group (n new tokens over the pruned prefix of
    group 0), bf16, all host cores; scaled to tokens/s of the full model by L / sample_layers."""
    from oracle import qp_oracle as O
    model, frames, fh, fw, gs, rho, prefix, tail = CONFIGS[name]
    ps = PRESETS[model]
    # torch-CPU on the GPU box's 256-thread EPYC host slows down past ~32 threads for these op sizes (measured:
    # tools/probe/cpu_diag.py), so the baseline uses min(host cores, 32) threads and reports that count.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    spec = O.TextSpec(hidden=ps.hidden, n_heads=ps.n_heads, n_kv_heads=ps.n_kv_heads, head_dim=ps.head_dim,
                      intermediate=ps.intermediate, n_layers=sample_layers, vocab=8)
    gh, gw = fh // 14, fw // 14
    n_video = (frames // 2) * (gh // 2) * (gw // 2)
    plan = planner.plan_groups(frames, gs, gh, gw, prefix, prefix + n_video + tail)
    n0, n1 = plan.tokens[0], min(sample_tokens, plan.tokens[min(1, len(plan.tokens) - 1)])
    P = int(n0 * rho)
    g = torch.Generator().manual_seed(0)
    rn = lambda *s, sc=0.02: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16)
    w = {}
    for l in range(sample_layers):
        p = f"layers.{l}."
        w[p + "input_layernorm.weight"] = torch.ones(spec.hidden, dtype=torch.bfloat16)
        w[p + "post_attention_layernorm.weight"] = torch.ones(spec.hidden, dtype=torch.bfloat16)
        w[p + "q_proj.weight"], w[p + "q_proj.bias"] = rn(spec.n_heads * 128, spec.hidden), rn(spec.n_heads * 128)
        w[p + "k_proj.weight"], w[p + "k_proj.bias"] = rn(spec.n_kv_heads * 128, spec.hidden), rn(spec.n_kv_heads * 128)
        w[p + "v_proj.weight"], w[p + "v_proj.bias"] = rn(spec.n_kv_heads * 128, spec.hidden), rn(spec.n_kv_heads * 128)
        w[p + "o_proj.weight"] = rn(spec.hidden, spec.n_heads * 128)
        w[p + "mlp.gate_proj.weight"], w[p + "mlp.up_proj.weight"] = rn(spec.intermediate, spec.hidden), rn(spec.intermediate, spec.hidden)
        w[p + "mlp.down_proj.weight"] = rn(spec.hidden, spec.intermediate)
    cache = O.OracleCache(sample_layers)
    for l in range(sample_layers):
        cache.append(l, rn(spec.n_kv_heads, P, 128, sc=1.0), rn(spec.n_kv_heads, P, 128, sc=1.0))
    h = rn(n1, spec.hidden, sc=0.5)
    pos = torch.arange(n1)[None].repeat(3, 1) + P
    cos, sin = O.mrope_cos_sin(pos, spec, torch.bfloat16)
    k_keep = O.effective_k(n1, None, rho, None, None, 0, ps.n_layers)
    t0 = time.perf_counter()
    with torch.no_grad():
        for l in range(sample_layers):
            h, _, cos, sin = O.decoder_layer(h, w, l, spec, cache, cos, sin, k_keep)
    dt = time.perf_counter() - t0
    tok_s = n1 / (dt / sample_layers * ps.n_layers)
    return {"value": round(tok_s, 2), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{sample_layers} of {ps.n_layers} decoder layers (bf16 torch-CPU oracle incl. key-norm prune) over the first "
                      f"{n1} new tokens of group 1 on a {P}-token pruned prefix, {dt:.2f}s, scaled by L/{sample_layers}"}


def pipeline_leg(name, eng, device):
    """video -> first token with the real front end: synthetic frame source (CPU producer thread) -> pinned ring -> H2D
    on a copy stream -> GPU normalise/patchify + ViT on a second stream -> group prefill -> tail -> first token id on
    the host.  Reported next to the headline number (which excludes the ViT, like SURVEY §8d 'with and without ViT')."""
    from quickvideo_amd.pipeline import PrefillPipeline, QwenVLNative
    from quickvideo_amd.processor import SyntheticProcessor
    from quickvideo_amd.vit import VisionWeights
    from quickvideo_amd.lvu import _VIT
    model, frames, fh, fw, gs, rho, prefix, tail = CONFIGS[name]
    vis = VisionWeights.synthetic(_VIT[model], device, seed=0)
    m = QwenVLNative(eng.w, vis, device, name=model)
    cfg = LVUConfig(model, top_p=rho, video_group_size=gs, num_frames=frames)
    pipe = PrefillPipeline(m, cfg, SyntheticProcessor(eng.spec), ops=eng.ops)
    video = f"synthetic://?frames={frames * 4}&h=1080&w=1920&fps=2&seed=1"
    res = {}
    for mode, overlap in (("overlapped", True), ("sequential", False)):
        pipe.generate("Describe what happens in this video in detail.", video, max_new_tokens=1, overlap=overlap)   # warm-up
        pipe.generate("Describe what happens in this video in detail.", video, max_new_tokens=1, overlap=overlap)
        t = pipe.last_timings
        res[mode] = {"ttft_ms": round(t.ttft * 1e3, 2), "frame_wait_ms": round(t.fetch * 1e3, 2), "vit_ms": round(t.vit * 1e3, 2),
                     "group_loop_ms": round(t.prefill * 1e3, 2), "prefill_tokens_per_s_with_vit": round(t.tokens / t.prefill, 1),
                     "tokens": t.tokens}
    res["note"] = "frames: seeded synthetic uint8 generated on the host CPU (no codec in the image); ViT: Qwen2-VL 32-layer tower, random weights"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the greedy-decode leg (hipGraph step, ms per token)")
    ap.add_argument("--no-ttft", action="store_true", help="skip the extra step that times prefill -> first token id on the host")
    ap.add_argument("--parallel", default="auto", choices=["auto", "sp", "tp", "pp"],
                    help="N>1: sp = group-token parallel (replicated weights/KV, one K/V all-gather per layer), "
                         "tp = tensor parallel over heads / MLP columns (two [n,d] all-reduces per layer), "
                         "pp = layer pipeline (each rank holds L/N layers and their KV; one [n,d] hand-off per group and stage); "
                         "auto = pp when the video has >= 4*N groups (the pipeline stays full), else sp")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    # QP_BENCH_SINGLE_DEVICE=1: developer hook to exercise the multi-process path on a 1-GPU box (all ranks on cuda:0, gloo
    # collectives) — never used by the driver, numbers from it are meaningless.
    single_dev = os.environ.get("QP_BENCH_SINGLE_DEVICE") == "1"
    if single_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    tp_group = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if single_dev:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=device)    # nccl == RCCL over xGMI on ROCm
        tp_group = torch.distributed.group.WORLD

    # layout = (pp stages) x (sp ranks per stage); auto picks it from the number of groups (long videos keep a layer pipeline full,
    # short ones split each group's tokens); "sp" / "pp" force the pure forms, "tp" is tensor parallel
    _, frames_, _, _, gs_, _, _, _ = CONFIGS[args.config]
    n_groups_ = -(-frames_ // gs_)
    layout = (1, 1)
    if world > 1:
        layout = {"auto": choose_layout(n_groups_, world), "sp": (1, world), "pp": (world, 1), "tp": (1, 1)}[args.parallel]
    if args.parallel != "tp":
        args.parallel = "single" if world == 1 else ("sp" if layout[0] == 1 else "pp" if layout[1] == 1 else "ppsp")
    spec, cfg, plan, eng, embeds, pos, T = build_workload(args.config, device, rank, world, parallel=args.parallel, layout=layout)
    if args.parallel == "tp":
        eng.tp_group = tp_group
    elif world > 1:
        pp_n, sp_n = layout
        if sp_n == world:
            eng.sp_group = tp_group
        elif sp_n > 1:                        # every rank creates every stage's group, in the same order
            stage_groups = [torch.distributed.new_group(ranks=list(range(s * sp_n, (s + 1) * sp_n))) for s in range(pp_n)]
            eng.sp_group = stage_groups[rank // sp_n]
    tokens = sum(plan.tokens)                 # tokens prefetched in the group loop (the reference's total_prefill span)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # the engine chooses its GEMM decompositions / hipBLASLt algorithms the first time a segment size shows up (one-off, like
    # building the weights): with --warmup 0 that first use must not land in the timed steps
    for _ in range(max(args.warmup, 1)):
        run_step(eng, plan, embeds, pos)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tok = run_step(eng, plan, embeds, pos)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3

    # TTFT of the prefill leg: one step ending with the first token id on the host
    barrier()
    first, ttft_ms = int(tok.item()), None
    if not args.no_ttft:
        t1 = time.perf_counter()
        first = int(run_step(eng, plan, embeds, pos).item())
        ttft_ms = (time.perf_counter() - t1) * 1e3

    decode = None
    if world == 1 and not args.no_decode:
        decode = decode_leg(eng, first)            # the engine holds the cache of the last step (prefill + tail)

    lin, att, prune_bytes = flops_and_bytes(spec, cfg, plan, world)
    roofline = None
    extra = {}
    if not args.no_kernel_timing:
        timed = TimedOps(eng.ops, ["prefill_attn", "prune_staged", "rope_append", "add_rmsnorm", "swiglu"])
        real_ops, eng.ops = eng.ops, timed
        run_step(eng, plan, embeds, pos)
        eng.ops = real_ops
        tot = timed.totals_ms()
        att_ms, att_n = tot["prefill_attn"]
        att_local = att / world                                  # tp: heads sharded
        if world > 1 and args.parallel != "tp":                  # rank 0: its stage's layers x its zigzag query rows
            pp_n, sp_n = layout
            l0_, l1_ = pp_layer_split(spec.n_layers, pp_n, rank // sp_n)
            att_local, Pp = 0.0, 0
            for n in plan.tokens:
                if sp_n > 1 and n >= 64 * sp_n:
                    for lo, hi in sp_row_ranges(n, sp_n, rank % sp_n):
                        att_local += 4.0 * spec.n_layers * spec.n_heads * spec.head_dim * sum(Pp + i + 1 for i in range(lo, hi))
                else:
                    att_local += spec.attn_flops(n, Pp)
                Pp += effective_k(n, cfg, 0, spec.n_layers) or n
            att_local = (att_local + spec.attn_flops(plan.tail_len, Pp)) * (l1_ - l0_) / spec.n_layers
        ach = att_local / (att_ms * 1e-3) / 1e12
        traffic = None            # HBM bytes per launch from the committed rocprofv3 PMC passes (tools/profile_bench.sh), same command
        tpath = os.path.join(ROOT, "profiles", "attn_pmc_traffic_latest.json")
        if args.config == "cfg2" and world == 1 and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("attn_fwd_kernel_s6", {}).get("traffic_bytes_per_launch")
        roofline = {"kernel": "attn_fwd_kernel_s6 (MFMA prefill attention over pruned prefix + causal tail)", "bound": "mfma",
                    "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                    "traffic": traffic, "launches": att_n, "avg_launch_ms": round(att_ms / max(att_n, 1), 4),
                    "algorithmic_flops_per_step": att_local}
        pr_ms = tot["prune_staged"][0]
        if pr_ms > 0:
            pb = prune_bytes / world if world > 1 else prune_bytes
            extra["roofline_prune"] = {"kernels": "prune_fused_kernel (select + gather, one launch)", "bound": "hbm",
                                       "achieved": round(pb / (pr_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": round(pb / (pr_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "launches": tot["prune_staged"][1],
                                       "ms_per_step": round(pr_ms, 3), "algorithmic_bytes_per_step": pb}
        extra["kernel_ms_per_step"] = {k: round(v[0], 3) for k, v in tot.items()}

    pipe_stats = None
    if world == 1 and not args.no_pipeline:
        pipe_stats = pipeline_leg(args.config, eng, device)

    if rank == 0:
        out = {
            "metric": "prefill_tokens_per_s", "value": round(tokens / (ms_per_step * 1e-3), 1), "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: {CONFIGS[args.config][0]}, {CONFIGS[args.config][1]} frames "
                                   f"{CONFIGS[args.config][2]}x{CONFIGS[args.config][3]}, group_size {CONFIGS[args.config][4]}, "
                                   f"key-norm rho={CONFIGS[args.config][5]}",
                       "groups": len(plan.tokens), "tokens_per_group": plan.tokens[-1], "prefill_tokens": tokens,
                       "tail_tokens": plan.tail_len, "layers": spec.n_layers, "parallelism": ("single" if world == 1 else f"tp{world}" if args.parallel == "tp" else
                                                                                   f"pp{layout[0]}xsp{layout[1]}"),
                       "vit": "excluded (synthetic ViT-output embeddings resident in HBM)",
                       "weights": "seeded random at real dims"},
            "ttft_ms_prefill_leg": None if ttft_ms is None else round(ttft_ms, 3), "first_token": first,
            "decode": decode,
            "algorithmic_tflop_per_step": round((lin + att) / 1e12, 2),
            "mfma_frac_whole_step": round((lin + att) / world / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "roofline": roofline,
        }
        out.update(extra)
        if pipe_stats:
            out["video_to_first_token"] = pipe_stats
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
