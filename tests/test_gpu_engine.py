"""GPU end-to-end parity: the engine on the HIP library vs (a) the reference's composite golden run (GV5),
(b) the CPU oracle on the same seeded inputs, at tiny and at real (7B) layer dimensions."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from oracle import qp_oracle as O
from oracle.make_golden import E2E_CASES
from quickvideo_amd import planner
from quickvideo_amd.engine import QuickPrefillEngine
from quickvideo_amd.lvu_config import LVUConfig
from quickvideo_amd.spec import TINY, QWEN2_VL_7B, TextSpec
from quickvideo_amd.weights import DecoderWeights
from tests.test_engine_host import make_case

pytestmark = pytest.mark.gpu

# Stated tolerance for bf16 end-to-end logits (|logit| ~ 1): GPU GEMM accumulation order differs from CPU.
ATOL, COS = 4e-2, 0.999


def run_gpu(spec, w, plan, pos, embeds, cfg):
    dw = DecoderWeights.from_named(spec, w, "cuda:0")
    eng = QuickPrefillEngine(dw, cfg, capacity=embeds.shape[0] + 8, max_group_tokens=max(plan.tokens + [plan.tail_len]), device="cuda:0")
    eng.kept_trace = []
    post = torch.from_numpy(pos).cuda()
    e = embeds.cuda()
    start = 0
    for n in plan.tokens:
        eng.prefill_group(e[start:start + n], post[:, start:start + n])
        start += n
    logits = eng.prefill_tail(e[start:], post[:, start:])
    torch.cuda.synchronize()
    return eng, logits.cpu()


def check_logits(got, ref):
    got, ref = np.asarray(got, dtype=np.float32), np.asarray(ref, dtype=np.float32)
    assert np.isfinite(got).all()
    assert np.max(np.abs(got - ref)) <= ATOL, np.max(np.abs(got - ref))
    assert float(np.dot(got, ref) / (np.linalg.norm(got) * np.linalg.norm(ref))) >= COS


@pytest.mark.parametrize("ci", [i for i, c in enumerate(E2E_CASES) if c[1] == "bfloat16"])
def test_engine_vs_reference_golden(golden_dir, ci):
    data = np.load(os.path.join(golden_dir, "gv5_e2e.npz"))
    name, dtn, frames, gh, gw, gs, prefix, tail, top_p, top_k = E2E_CASES[ci]
    spec_o, w, plan, pos, delta, embeds = make_case(frames, gh, gw, gs, prefix, tail)
    eng, logits = run_gpu(TINY, w, plan, pos, embeds, LVUConfig("x", top_p=top_p, top_k=top_k, video_group_size=gs))
    assert eng.arena.len == list(data[f"{name}_cache_len"])
    check_logits(logits.numpy(), data[f"{name}_logits"])


@pytest.mark.parametrize("top_p,pps", [(0.5, None), (0.25, None), (0.5, 1), (None, None)])
def test_engine_vs_oracle_tiny(top_p, pps):
    spec_o, w, plan, pos, delta, embeds = make_case(24, 12, 16, 8, 15, 20)      # 3 groups x 192 tokens
    cfg = LVUConfig("x", top_p=top_p, prefill_prune_starting_layer=pps, video_group_size=8)
    eng, logits = run_gpu(TINY, w, plan, pos, embeds, cfg)
    ref = O.group_prefill(w, spec_o, embeds, pos, plan.tokens, O.PruneCfg(top_p=top_p, prefill_prune_starting_layer=pps))
    assert eng.arena.len == ref["cache_len"]
    check_logits(logits.numpy(), ref["logits"].numpy())
    # kept sets: identical up to near-ties created by GEMM rounding differences in K — report overlap, require >= 90 %
    flat_ref = [k for g in ref["kept"] for k in g]
    tot = same = 0
    for (l, got), want in zip(eng.kept_trace, flat_ref):
        assert (got is None) == (want is None)
        if want is not None:
            g = got.cpu().numpy()
            assert len(g) == len(want) and np.all(np.diff(g) > 0)
            tot += len(want); same += len(set(g.tolist()) & set(want.tolist()))
    if tot:
        assert same / tot >= 0.90, same / tot


def test_engine_many_groups_vs_oracle():
    """24 sequential groups (the 1-hour video has 450): the pruned prefix grows group after group, every group attends over what the
    earlier prunes kept.  Cache lengths exact after every group, logits within the stated tolerance, kept sets >= 90 % identical."""
    spec_o, w, plan, pos, delta, embeds = make_case(192, 8, 8, 8, 15, 20)      # 24 groups x 64 tokens (+15 prefix on group 0)
    assert len(plan.tokens) == 24
    cfg = LVUConfig("x", top_p=0.5, video_group_size=8)
    eng, logits = run_gpu(TINY, w, plan, pos, embeds, cfg)
    ref = O.group_prefill(w, spec_o, embeds, pos, plan.tokens, O.PruneCfg(top_p=0.5))
    assert eng.arena.len == ref["cache_len"]
    check_logits(logits.numpy(), ref["logits"].numpy())
    flat_ref = [k for g in ref["kept"] for k in g]
    assert len(eng.kept_trace) == len(flat_ref)
    tot = same = 0
    for (l, got), want in zip(eng.kept_trace, flat_ref):
        if want is not None:
            g = got.cpu().numpy()
            assert len(g) == len(want) and np.all(np.diff(g) > 0)
            tot += len(want); same += len(set(g.tolist()) & set(want.tolist()))
    assert same / tot >= 0.90, same / tot


def test_engine_real_dims_one_layer():
    """One decoder layer at Qwen2-VL-7B dimensions (d=3584, 28/4 heads, I=18944), 2 groups of 320 + tail."""
    spec = TextSpec(hidden=3584, n_heads=28, n_kv_heads=4, head_dim=128, intermediate=18944, n_layers=1, vocab=1024)
    spec_o = O.TextSpec(hidden=3584, n_heads=28, n_kv_heads=4, head_dim=128, intermediate=18944, n_layers=1, vocab=1024)
    w = {k: v.to(torch.bfloat16) for k, v in O.synthetic_text_weights(spec_o, seed=11, norm_jitter=0.05).items()}
    frames, gh, gw, gs, prefix, tail = 8, 16, 20, 4, 15, 24        # 4 frame pairs x 80 tokens
    n_video = (frames // 2) * (gh // 2) * (gw // 2)
    T = prefix + n_video + tail
    plan = planner.plan_groups(frames, gs, gh, gw, prefix, T)
    pos, _ = planner.mrope_positions(prefix, (frames // 2, gh, gw), tail)
    rs = np.random.RandomState(5)
    embeds = torch.from_numpy(rs.standard_normal((T, spec.hidden)).astype(np.float32) * 0.5).to(torch.bfloat16)
    eng, logits = run_gpu(spec, w, plan, pos, embeds, LVUConfig("x", top_p=0.5, video_group_size=gs))
    ref = O.group_prefill(w, spec_o, embeds, pos, plan.tokens, O.PruneCfg(top_p=0.5))
    assert eng.arena.len == ref["cache_len"]
    check_logits(logits.numpy(), ref["logits"].numpy())


def test_engine_local_attention_off_vs_oracle():
    """adaptive_local_attention=False on the GPU (qwen25_lvu.py:700-714): every video group is prefilled WITHOUT the earlier groups'
    K/V (their pruned rows still accumulate in the cache), the prompt tail attends to all of it.  Oracle: each group through an empty
    cache, caches concatenated, then the tail — exactly what the reference's branch does."""
    spec_o, w, plan, pos, delta, embeds = make_case(24, 12, 16, 8, 15, 20)      # 3 groups x 192 tokens
    cfg = LVUConfig("x", top_p=0.5, video_group_size=8, adaptive_local_attention=False)
    eng, logits = run_gpu(TINY, w, plan, pos, embeds, cfg)
    # oracle: groups independently
    post = torch.from_numpy(pos)
    caches, start = [], 0
    for n in plan.tokens:
        r = O.group_prefill(w, spec_o, embeds[start:start + n + 1], pos[:, start:start + n + 1], [n], O.PruneCfg(top_p=0.5), want_logits=False)
        c = r["cache"]
        for l in range(spec_o.n_layers):                       # drop the 1-token dummy tail group_prefill appended
            c.k[l], c.v[l] = c.k[l][:, :-1], c.v[l][:, :-1]
        caches.append(c)
        start += n
    merged = O.OracleCache(spec_o.n_layers)
    for c in caches:
        for l in range(spec_o.n_layers):
            merged.append(l, c.k[l], c.v[l])
    h = embeds[start:]
    cos, sin = O.mrope_cos_sin(post[:, start:], spec_o, embeds.dtype)
    for l in range(spec_o.n_layers):
        h, _, cos, sin = O.decoder_layer(h, w, l, spec_o, merged, cos, sin, None)
    assert eng.arena.len == [merged.length(l) for l in range(spec_o.n_layers)]
    ref = torch.nn.functional.linear(O.rmsnorm(h[-1:], w["norm.weight"], spec_o.rms_eps), w["lm_head.weight"])[0].float()
    check_logits(logits.numpy(), ref.numpy())
    # and it differs from the default (cross-group attention on): the switch is not a no-op
    eng2, logits2 = run_gpu(TINY, w, plan, pos, embeds, LVUConfig("x", top_p=0.5, video_group_size=8))
    assert float((logits - logits2).abs().max()) > 1e-3


def test_engine_72b_tp8_rank_slice_vs_oracle():
    """cfg5's per-rank problem (Qwen2-VL-72B under TP=8: d=8192, 8 q heads + 1 kv head per GPU, I/8 = 3696 MLP columns), two layers,
    through forward_segment: 2 groups of 480 tokens + tail vs the oracle on a model with exactly those dims (the rank's partial sums
    are what the all-reduce would add; here the single 'rank' is the whole model, so the outputs are comparable as they are)."""
    dims = dict(hidden=8192, n_heads=8, n_kv_heads=1, head_dim=128, intermediate=3696, n_layers=2, vocab=1024)
    spec, spec_o = TextSpec(**dims), O.TextSpec(**dims)
    # std 0.01: at d = 8192 the usual 0.02 gives q.k/sqrt(D) a std of ~3.3, i.e. attention so peaky that ONE token moving across the
    # prune threshold (2 of 240 kept tokens differ between hipBLASLt's and the CPU's K rounding) moves the logits by 0.6
    # (tools/probe/dbg_engine_slice.py); with 0.01 the same two differences move them by 0.02
    w = O.hashed_text_weights(spec_o, seed=21, device="cuda", norm_jitter=0.05, std=0.01)
    frames, gh, gw, gs, prefix, tail = 16, 16, 30, 8, 15, 24       # 8 frame pairs x 120 tokens -> 2 groups of 480
    T = prefix + (frames // 2) * (gh // 2) * (gw // 2) + tail
    plan = planner.plan_groups(frames, gs, gh, gw, prefix, T)
    pos, _ = planner.mrope_positions(prefix, (frames // 2, gh, gw), tail)
    embeds = O.hashed_normal((T, spec.hidden), 22, 0.5)
    eng, logits = run_gpu(spec, w, plan, pos, embeds, LVUConfig("x", top_p=0.5, video_group_size=gs))
    ref = O.group_prefill({k: v.cpu() for k, v in w.items()}, spec_o, embeds, pos, plan.tokens, O.PruneCfg(top_p=0.5))
    assert eng.arena.len == ref["cache_len"]
    check_logits(logits.numpy(), ref["logits"].numpy())


def _oracle_trace(w, spec_o, embeds, pos, group_tokens, top_p):
    """oracle.group_prefill's loop with the residual stream recorded after every layer: -> (hidden[(segment, layer)] fp32, kept, cache_len)."""
    cache = O.OracleCache(spec_o.n_layers)
    post, start, hid, kept = torch.from_numpy(pos), 0, {}, {}
    segs = list(group_tokens) + [embeds.shape[0] - sum(group_tokens)]
    for gi, n in enumerate(segs):
        tail = gi == len(segs) - 1
        h = embeds[start:start + n]
        cos, sin = O.mrope_cos_sin(post[:, start:start + n], spec_o, embeds.dtype)
        for l in range(spec_o.n_layers):
            kk = None if tail else O.effective_k(n, None, top_p, None, None, l, spec_o.n_layers)
            h, kp, cos, sin = O.decoder_layer(h, w, l, spec_o, cache, cos, sin, kk)
            hid[(gi, l)], kept[(gi, l)] = h.float(), kp
        start += n
    return hid, kept, [cache.length(l) for l in range(spec_o.n_layers)]


def _row_cosine_min_mean(a, b):
    c = torch.nn.functional.cosine_similarity(a.float(), b.float(), dim=-1)
    return float(c.min()), float(c.mean())


def test_engine_cfg5_rank_shape_vs_oracle(monkeypatch):
    """BASELINE.json configs[4] (Qwen2-VL-72B, TP=8, 512 frames of 224x420, group_size 16, rho 0.5) at its OWN per-rank shape:
    d = 8192, 8 q heads + 1 kv head, I/8 = 3696 MLP columns, groups of n = 960 tokens (8 frame pairs x 120), k = 480, 4 layers,
    DEFAULT-std (0.02) hash-generated weights; 3 groups + prompt tail through forward_segment on the GPU vs the CPU oracle.

    At this width attention is peaky (q.k/sqrt(D) has std ~3) and final logits amplify a handful of near-tie flips in the kept
    sets (two implementations that are both right differ by 0.6 in a logit: tools/probe/dbg_engine_slice.py), so the logit check
    lives in test_engine_72b_tp8_rank_slice_vs_oracle with damped weights.  Here the ROBUST quantities are pinned, and the bar is
    not "what the GPU produced" but the distance between TWO CPU evaluations of the same model that differ only in GEMM
    accumulation (torch's bf16 linear vs fp32-accumulated matmul rounded once): the GPU may be at most twice as far from the
    oracle as that second CPU implementation is, plus a stated absolute slack —
        cache lengths: exact;
        kept-set overlap per (group, layer): 1 - ov_gpu <= 2 * (1 - ov_cpu2) + 0.02;
        residual-stream rows after every layer, mean row cosine: 1 - cos_gpu <= 2 * (1 - cos_cpu2) + 2e-3; worst row >= 0.90."""
    dims = dict(hidden=8192, n_heads=8, n_kv_heads=1, head_dim=128, intermediate=3696, n_layers=4, vocab=1024)
    spec, spec_o = TextSpec(**dims), O.TextSpec(**dims)
    w = O.hashed_text_weights(spec_o, seed=31, device="cuda", norm_jitter=0.05)            # std = 0.02, the synthetic default
    frames, gh, gw, gs, prefix, tail = 48, 16, 30, 16, 15, 24                            # 224x420 frames -> 16x30 patches; 3 groups x 960
    T = prefix + (frames // 2) * (gh // 2) * (gw // 2) + tail
    plan = planner.plan_groups(frames, gs, gh, gw, prefix, T)
    assert plan.tokens == [975, 960, 960]
    pos, _ = planner.mrope_positions(prefix, (frames // 2, gh, gw), tail)
    embeds = O.hashed_normal((T, spec.hidden), 32, 0.5)
    cfg = LVUConfig("x", top_p=0.5, video_group_size=gs)
    dw = DecoderWeights.from_named(spec, w, "cuda:0")
    eng = QuickPrefillEngine(dw, cfg, capacity=T + 8, max_group_tokens=max(plan.tokens + [plan.tail_len]), device="cuda:0")
    eng.kept_trace, eng.hidden_trace = [], []
    post, e, start = torch.from_numpy(pos).cuda(), embeds.cuda(), 0
    for n in plan.tokens:
        eng.prefill_group(e[start:start + n], post[:, start:start + n]); start += n
    eng.prefill_tail(e[start:], post[:, start:])
    torch.cuda.synchronize()
    wc = {k: v.cpu() for k, v in w.items()}
    hid, kept, clen = _oracle_trace(wc, spec_o, embeds, pos, plan.tokens, 0.5)
    # second CPU implementation: same oracle code, every linear accumulated in fp32 and rounded once
    lin = lambda x, ww, b=None: ((x.float() @ ww.float().t()) + (0 if b is None else b.float())).to(x.dtype)    # noqa: E731
    monkeypatch.setattr(torch.nn.functional, "linear", lin)
    hid2, kept2, clen2 = _oracle_trace(wc, spec_o, embeds, pos, plan.tokens, 0.5)
    monkeypatch.undo()
    assert eng.arena.len == clen == clen2
    L, S = spec.n_layers, len(plan.tokens) + 1
    assert len(eng.hidden_trace) == S * L and len(eng.kept_trace) == S * L
    rows = []
    for si in range(S):
        for l in range(L):
            (lg, got), (lh, hg) = eng.kept_trace[si * L + l], eng.hidden_trace[si * L + l]
            assert lg == l and lh == l
            cmin, cmean = _row_cosine_min_mean(hg.cpu(), hid[(si, l)])
            _, cmean2 = _row_cosine_min_mean(hid2[(si, l)], hid[(si, l)])
            ov = ov2 = 1.0
            if kept[(si, l)] is not None:
                g, want, w2 = got.cpu().numpy(), kept[(si, l)], kept2[(si, l)]
                assert len(g) == len(want) and np.all(np.diff(g) > 0)
                ov = len(set(g.tolist()) & set(want.tolist())) / len(want)
                ov2 = len(set(w2.tolist()) & set(want.tolist())) / len(want)
            rows.append((si, l, ov, ov2, cmean, cmean2, cmin))
    print("segment layer  overlap(gpu) overlap(cpu2)  mean-row-cos(gpu) (cpu2)  min-row-cos(gpu)")
    for r in rows:
        print("   %d      %d      %.4f       %.4f          %.5f      %.5f     %.4f" % r)
    for si, l, ov, ov2, cmean, cmean2, cmin in rows:
        assert 1 - ov <= 2 * (1 - ov2) + 0.02, (si, l, ov, ov2)
        assert 1 - cmean <= 2 * (1 - cmean2) + 2e-3, (si, l, cmean, cmean2)
        assert cmin >= 0.90, (si, l, cmin)


@pytest.mark.parametrize("pt", ["query_attention_weights", "query_attention_weights_by_value_norm"])
def test_engine_query_based_vs_oracle(pt):
    """SURVEY 8 f4 on the GPU: prompt-appended groups, qp_query_scores + qp_prune_keys, the shifted causal alignment as two attention
    launches — vs the oracle's restatement of the reference's query-based mode (pinned op by op on GV9)."""
    spec_o, w, plan, pos, delta, embeds = make_case(24, 12, 16, 8, 15, 20)
    m = plan.tail_len
    cfg = LVUConfig("x", top_p=0.5, video_group_size=8, top_k_predict_type=pt)
    dw = DecoderWeights.from_named(TINY, w, "cuda:0")
    eng = QuickPrefillEngine(dw, cfg, capacity=embeds.shape[0] + 8 + m, max_group_tokens=max(plan.tokens) + m, device="cuda:0")
    eng.kept_trace = []
    post, e, start = torch.from_numpy(pos).cuda(), embeds.cuda(), 0
    for n in plan.tokens:
        eng.prefill_group(e[start:start + n], post[:, start:start + n + m], prompt_embeds=e[-m:])
        start += n
    logits = eng.prefill_tail(e[start:], post[:, start:]).cpu()
    ref = O.group_prefill(w, spec_o, embeds, pos, plan.tokens, O.PruneCfg(top_p=0.5, top_k_predict_type=pt))
    assert eng.arena.len == ref["cache_len"]
    check_logits(logits.numpy(), ref["logits"].numpy())
    flat = [k for g in ref["kept"] for k in g]
    tot = same = 0
    for (l, got), want in zip(eng.kept_trace, flat):
        if want is not None:
            g = got.cpu().numpy()
            assert len(g) == len(want) and np.all(np.diff(g) > 0)
            tot += len(want); same += len(set(g.tolist()) & set(want.tolist()))
    assert same / tot >= 0.9, same / tot


def test_engine_query_based_large_group_vs_oracle():
    """Query-score pruning on a group of more than 8192 tokens (the single-group / large-group settings of the reference,
    `video_group_size` up to the whole video): the score keys go through qp_select_keys + qp_gather_kv instead of the one-launch
    prune.  8448-token group, tiny dims, vs the oracle's restatement of LVUCache.update's scoring."""
    pt = "query_attention_weights"
    spec_o, w, plan, pos, delta, embeds = make_case(48, 32, 44, 48, 15, 20)            # ONE group of 24 x 352 = 8448 (+15) tokens
    assert max(plan.tokens) > 8192
    m = plan.tail_len
    cfg = LVUConfig("x", top_p=0.5, video_group_size=48, top_k_predict_type=pt)
    dw = DecoderWeights.from_named(TINY, w, "cuda:0")
    eng = QuickPrefillEngine(dw, cfg, capacity=embeds.shape[0] + 8 + m, max_group_tokens=max(plan.tokens) + m, device="cuda:0")
    eng.kept_trace = []
    post, e, start = torch.from_numpy(pos).cuda(), embeds.cuda(), 0
    for n in plan.tokens:
        eng.prefill_group(e[start:start + n], post[:, start:start + n + m], prompt_embeds=e[-m:])
        start += n
    logits = eng.prefill_tail(e[start:], post[:, start:]).cpu()
    ref = O.group_prefill(w, spec_o, embeds, pos, plan.tokens, O.PruneCfg(top_p=0.5, top_k_predict_type=pt))
    assert eng.arena.len == ref["cache_len"]
    check_logits(logits.numpy(), ref["logits"].numpy())
    flat = [k for g in ref["kept"] for k in g]
    tot = same = 0
    for (l, got), want in zip(eng.kept_trace, flat):
        if want is not None:
            g = got.cpu().numpy()
            assert len(g) == len(want) and np.all(np.diff(g) > 0)
            tot += len(want); same += len(set(g.tolist()) & set(want.tolist()))
    assert same / tot >= 0.9, same / tot


def test_engine_cfg3_shape_two_layers_vs_oracle():
    """BASELINE.json configs[2] as a whole-engine run at the real layer width: groups of 2880 tokens (32 frames of 280x504), rho = 0.25
    (k = 720), three groups + tail through two 7B-dim layers vs the oracle: cache lengths exact, logits within tolerance, kept sets."""
    dims = dict(hidden=3584, n_heads=28, n_kv_heads=4, head_dim=128, intermediate=18944, n_layers=2, vocab=2048)
    spec, spec_o = TextSpec(**dims), O.TextSpec(**dims)
    w = O.hashed_text_weights(spec_o, seed=31, device="cuda", norm_jitter=0.05)
    frames, gh, gw, gs, prefix, tail = 96, 20, 36, 32, 15, 30         # 280x504 -> 20x36 patches -> 180 tokens per frame pair; 3 groups
    T = prefix + (frames // 2) * (gh // 2) * (gw // 2) + tail
    plan = planner.plan_groups(frames, gs, gh, gw, prefix, T)
    assert plan.tokens == [2895, 2880, 2880]
    pos, _ = planner.mrope_positions(prefix, (frames // 2, gh, gw), tail)
    embeds = O.hashed_normal((T, spec.hidden), 32, 0.5)
    eng, logits = run_gpu(spec, w, plan, pos, embeds, LVUConfig("x", top_p=0.25, video_group_size=gs))
    ref = O.group_prefill({k: v.cpu() for k, v in w.items()}, spec_o, embeds, pos, plan.tokens, O.PruneCfg(top_p=0.25))
    assert eng.arena.len == ref["cache_len"] == [723 + 720 + 720 + 30] * 2
    check_logits(logits.numpy(), ref["logits"].numpy())
    flat = [k for g in ref["kept"] for k in g]
    tot = same = 0
    for (l, got), want in zip(eng.kept_trace, flat):
        if want is not None:
            tot += len(want); same += len(set(got.cpu().numpy().tolist()) & set(want.tolist()))
    assert same / tot >= 0.97, same / tot


def test_pipeline_ring_reuse_under_a_slow_main_stream():
    """Many groups through the overlapped pipeline while the main stream is artificially slow (a spin kernel in front of every group's
    prefill), so the producer and the ViT stream run as far ahead as the 3-slot ring lets them — the situation in which a ring slot or a
    ViT output block could be recycled before the main stream has read it if the explicit orderings (read-done events handed back to the
    producer, record_stream on the ViT output) were missing.  The tokens must equal the everything-fetched-first run.  (Honest note: with
    the orderings disabled this tiny-model run still passed on the test box — the caching allocator did not happen to reuse the block —
    so the test guards against deadlock / mis-ordering under skew, it is not a proof of the race.)"""
    import lvu
    from quickvideo_amd.engine import QuickPrefillEngine as E
    from quickvideo_amd.lvu import load_native_model
    from quickvideo_amd.pipeline import PrefillPipeline
    from quickvideo_amd.processor import SyntheticProcessor
    m = load_native_model("synthetic:tiny", device="cuda:0", seed=5)
    # 16 frames of 448x672 per group: 3072 tokens -> 1.5 MB feature blocks, the size class of the ViT's own activations (large pool)
    cfg = lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=16, num_frames=192, extra_kwargs={"max_pixels": 448 * 672})   # 12 groups
    video = "synthetic://?frames=800&h=448&w=672&seed=9"
    pipe = PrefillPipeline(m, cfg, SyntheticProcessor(m.spec))
    ref = pipe.generate("What is shown?", video, max_new_tokens=4, overlap=False)
    orig = E.prefill_group

    def slow(self, *a, **kw):
        torch.cuda._sleep(40_000_000)                                  # ~20 ms of GPU spin on the main stream
        return orig(self, *a, **kw)
    E.prefill_group = slow
    try:
        for _ in range(2):
            got = pipe.generate("What is shown?", video, max_new_tokens=4, overlap=True)
            assert got == ref
    finally:
        E.prefill_group = orig
    assert pipe.last_timings.groups == 12


def test_no_gpu_fallback_is_loud():
    """The product refuses to run without the HIP library (no silent CPU path)."""
    from quickvideo_amd import native
    with pytest.raises(native.QuickPrefillUnavailable):
        native.load_library("/nonexistent/libquickprefill.so")


def test_lvu_generate_on_gpu(capsys):
    """Drop-in API on the GPU: frames -> H2D ring -> ViT -> group prefill -> tokens, overlapped and sequential agree."""
    import lvu
    from quickvideo_amd.lvu import load_native_model
    m = load_native_model("synthetic:tiny", device="cuda:0", seed=3)
    video = "synthetic://?frames=48&h=112&w=168&seed=2&pattern=gradient"
    outs = []
    for mt in ("qwen2vl_mi355x", "qwen2vl_mi355x_sequential"):
        obj = lvu.LVU(lvu.LVUConfig("synthetic:tiny", model_type=mt, top_p=0.5, video_group_size=4, num_frames=16), model=m)
        outs.append(obj.generate("What is shown?", video, max_new_tokens=4))
        t = obj._pipeline.last_timings
        assert t.groups == 4 and t.ttft > 0 and t.vit_span > 0 and t.gpu_prefill_busy > 0
    assert outs[0] == outs[1] and outs[0][0].count("<tok_") == 4
    assert "total time spent on prefill was" in capsys.readouterr().out


def test_generation_kwargs_on_gpu(monkeypatch):
    """HF-style generation kwargs on the GPU: the hipGraph decoder fed token by token (logits back to the selector each step) gives the
    same continuation as the per-op decode path; seeded sampling is reproducible; top_k=1 sampling is greedy."""
    import lvu
    from quickvideo_amd.decode import GraphDecoder
    from quickvideo_amd.lvu import load_native_model
    m = load_native_model("synthetic:tiny", device="cuda:0", seed=3)
    video = "synthetic://?frames=48&h=112&w=168&seed=2&pattern=gradient"
    obj = lvu.LVU(lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=4, num_frames=16), model=m)
    run = lambda **kw: obj.generate("What is shown?", video, max_new_tokens=8, eos_token_id=None, **kw)[0]
    greedy = run()
    assert run(do_sample=True, top_k=1) == greedy
    pen_graph = run(repetition_penalty=1.3)
    samp_graph = run(do_sample=True, temperature=3.0, top_p=0.9, seed=5)
    assert samp_graph == run(do_sample=True, temperature=3.0, top_p=0.9, seed=5) and samp_graph.count("<tok_") == 8
    monkeypatch.setattr(GraphDecoder, "supported", staticmethod(lambda eng: False))
    assert run(repetition_penalty=1.3) == pen_graph and run() == greedy


@pytest.mark.parametrize("mode", ["key_norms", "vector_norms", "vector_norms_small"])
def test_engine_other_norm_modes_vs_oracle(mode):
    """prune_mode through the engine: the other norm-based predict types against the composite oracle."""
    spec_o, w, plan, pos, delta, embeds = make_case(24, 12, 16, 8, 15, 20)
    cfg = LVUConfig("x", top_p=0.5, video_group_size=8, top_k_predict_type=mode)
    eng, logits = run_gpu(TINY, w, plan, pos, embeds, cfg)
    ref = O.group_prefill(w, spec_o, embeds, pos, plan.tokens, O.PruneCfg(top_p=0.5, top_k_predict_type=mode))
    assert eng.arena.len == ref["cache_len"]
    check_logits(logits.numpy(), ref["logits"].numpy())
    flat_ref = [k for g in ref["kept"] for k in g]
    tot = same = 0
    for (l, got), want in zip(eng.kept_trace, flat_ref):
        if want is not None:
            g = got.cpu().numpy()
            assert len(g) == len(want) and np.all(np.diff(g) > 0)
            tot += len(want); same += len(set(g.tolist()) & set(want.tolist()))
    assert same / tot >= 0.90, same / tot


def test_gemm_tuning_does_not_change_results(monkeypatch):
    """The warm-up tuner only chooses between decompositions / hipBLASLt algorithms of the same projections: with and without it
    (QP_TUNE_GEMMS) the engine must give the same cache lengths and logits within the end-to-end tolerance, at 7B width."""
    spec = TextSpec(hidden=3584, n_heads=28, n_kv_heads=4, head_dim=128, intermediate=18944, n_layers=1, vocab=1024)
    spec_o = O.TextSpec(hidden=3584, n_heads=28, n_kv_heads=4, head_dim=128, intermediate=18944, n_layers=1, vocab=1024)
    w = {k: v.to(torch.bfloat16) for k, v in O.synthetic_text_weights(spec_o, seed=21, norm_jitter=0.05).items()}
    frames, gh, gw, gs, prefix, tail = 8, 32, 40, 4, 15, 24        # 2 groups of 640 (+15) tokens: above the tuner's 256-row floor
    n_video = (frames // 2) * (gh // 2) * (gw // 2)
    T = prefix + n_video + tail
    plan = planner.plan_groups(frames, gs, gh, gw, prefix, T)
    pos, _ = planner.mrope_positions(prefix, (frames // 2, gh, gw), tail)
    embeds = torch.from_numpy(np.random.RandomState(6).standard_normal((T, spec.hidden)).astype(np.float32) * 0.5).to(torch.bfloat16)
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("QP_TUNE_GEMMS", flag)
        eng, logits = run_gpu(spec, w, plan, pos, embeds, LVUConfig("x", top_p=0.5, video_group_size=gs))
        outs.append((list(eng.arena.len), logits.numpy()))
    assert outs[0][0] == outs[1][0]
    check_logits(outs[1][1], outs[0][1])


def test_one_call_segment_path_equals_the_per_operator_loop(monkeypatch):
    """qp_prefill_segment (round 4): one library call per segment sequencing the SAME launches the per-operator loop issues.  With the same
    GEMM path on both sides (QP_GEMM_BACKEND=lt: every projection through qp_linear_act's plan for its shape, no row splits on either
    side because QP_TUNE_GEMMS=0) the two must agree BIT FOR BIT, first try, no retry: kept lists of every (group, layer), cache lengths, the cache rows
    themselves and the first-token logits; also under adaptive_local_attention=False, decay (different k per layer) and with layers that do
    not prune (rho = 1).  And the default configuration (its own tuned decompositions) stays within the engine's stated tolerance of the
    oracle (that is what every other test of this file now exercises)."""
    monkeypatch.setenv("QP_GEMM_BACKEND", "lt")
    monkeypatch.setenv("QP_TUNE_GEMMS", "0")
    spec_o, w, plan, pos, delta, embeds = make_case(24, 16, 24, 8, 15, 20)                  # 3 groups x 384 tokens
    for kw in (dict(top_p=0.5), dict(top_p=0.5, adaptive_local_attention=False), dict(top_p=0.5, top_k_decay_type="linear", top_k_decay_factor=0.5),
               dict(top_p=None)):
        cfg = LVUConfig("x", video_group_size=8, **kw)
        def one_run(native):
            monkeypatch.setenv("QP_NATIVE_SEGMENT", native)
            eng, logits = run_gpu(TINY, w, plan, pos, embeds, cfg)
            assert (eng._native_state is not None) == (native == "1")
            kept = [None if k is None else k.cpu().numpy() for _, k in eng.kept_trace]
            rows = [eng.arena.k(l)[:, :eng.arena.len[l]].cpu().view(torch.int16).numpy().copy() for l in range(3)]
            return list(eng.arena.len), kept, rows, logits.numpy().copy()

        # ONE comparison (rounds 3-4 retried here: one suite run in ~15 saw the cache rows differ by rounding with equal kept lists).  Root
        # cause, fixed in round 5 (csrc/qp_linear.hip: process-wide choice table): each engine owns a qp_ctx, the host-side "this GEMM shape is
        # tuned" table is per process — so the SECOND engine skipped qp_linear_tune and its fresh context ran hipBLASLt's candidate 0,
        # a different fp32 accumulation order whenever the first engine's stopwatch had picked another candidate (noise-dominated on these
        # tiny shapes).  test_tuned_gemm_choice_is_shared_by_every_context_of_the_process pins the mechanism;
        # tools/probe/repro_ctx_tuner_mismatch.py shows it on the round-4 library (profiles/r5_ctx_tuner_mismatch_repro.json).
        r_native, r_perop = one_run("1"), one_run("0")
        (l1, k1, r1, g1), (l0, k0, r0, g0) = r_native, r_perop
        assert l1 == l0, kw
        assert len(k1) == len(k0) and all((a is None) == (b is None) and (a is None or np.array_equal(a, b)) for a, b in zip(k1, k0)), kw
        bad = [(l, int((a != b).sum()), int(np.max(np.abs(a.astype(np.int32) - b.astype(np.int32)))), np.unique(np.argwhere(a != b)[:, 1])[:8].tolist())
               for l, (a, b) in enumerate(zip(r1, r0)) if not np.array_equal(a, b)]
        assert not bad, (kw, "K cache rows differ: (layer, elements, max bit distance, first rows)", bad, "logits max diff", float(np.max(np.abs(g1 - g0))))
        assert np.array_equal(g1, g0), (kw, float(np.max(np.abs(g1 - g0))))


def test_tuned_gemm_choice_is_shared_by_every_context_of_the_process():
    """The mechanism behind round 4's "flaky" one-call / per-operator mismatch: a GEMM problem tuned through one qp_ctx must run the SAME
    hipBLASLt candidate in every other context of the device — one created later (fresh plans) and one that had already planned the
    problem with the default pick — and give the same bits.  qp_linear_tune from the second context adopts the recorded pick (no second
    stopwatch run, which could disagree within timing noise)."""
    from quickvideo_amd.native import QuickPrefillOps
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    differ_from_default = 0
    for m, n, k in ((2233, 3584, 18944), (393, 384, 256), (27, 3584, 18944), (953, 3584, 3584)):      # row counts no other test uses
        x = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        ws = [(torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev) for _ in range(3)]
        early = QuickPrefillOps(dev)
        o_early = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        early.linear_act(x, ws[0], None, o_early, early.ACT_NONE)                # planned before anyone tuned: heuristic candidate 0
        assert early.linear_plan_choice(m, n, k) == (0, False)
        default_bits = o_early.view(torch.int16).clone()
        a = QuickPrefillOps(dev)
        assert a.linear_plan_choice(m, n, k) == (-1, False)
        o_a = torch.empty_like(o_early)
        a.linear_tune(x, ws, None, o_a)
        choice, tuned = a.linear_plan_choice(m, n, k)
        assert tuned and choice >= 0
        a.linear_act(x, ws[0], None, o_a, a.ACT_NONE)
        b = QuickPrefillOps(dev)                                                 # a later engine's context: never tunes (host table says "tuned")
        o_b = torch.empty_like(o_early)
        b.linear_act(x, ws[0], None, o_b, b.ACT_NONE)
        assert b.linear_plan_choice(m, n, k) == (choice, True)
        early.linear_act(x, ws[0], None, o_early, early.ACT_NONE)                # the early context converges on the recorded pick
        assert early.linear_plan_choice(m, n, k) == (choice, True)
        b.linear_tune(x, ws, None, o_b.clone())                                  # adopting, not re-timing
        assert b.linear_plan_choice(m, n, k) == (choice, True)
        torch.cuda.synchronize()
        assert torch.equal(o_a.view(torch.int16), o_b.view(torch.int16)) and torch.equal(o_a.view(torch.int16), o_early.view(torch.int16)), (m, n, k, choice)
        differ_from_default += int(choice != 0 and not torch.equal(default_bits, o_a.view(torch.int16)))
        # a context that plans under ANOTHER workspace limit gets another candidate list; the record is the algorithm itself,
        # so it either finds that very algorithm in its list (whatever its index there) and computes the same bits, or it does not hold it
        # and says so (tuned == 1, not 2) — then its own qp_linear_tune re-times, replaces the record, and the others converge on it
        c = QuickPrefillOps(dev)
        c._lt_ws[c._stream()] = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
        o_c = torch.empty_like(o_early)
        c.linear_act(x, ws[0], None, o_c, c.ACT_NONE)
        kind = ctypes.c_int32(0)
        idx = ctypes.c_int32(-1)
        assert c.lib.qp_linear_plan_choice(c.ctx, m, n, k, 0, 0, ctypes.byref(idx), ctypes.byref(kind)) == 0 and idx.value >= 0
        torch.cuda.synchronize()
        if kind.value == 2:
            assert torch.equal(o_c.view(torch.int16), o_a.view(torch.int16)), (m, n, k, "same algorithm, different bits")
        else:
            assert kind.value == 1
            c.linear_tune(x, ws, None, o_c)                                       # not in its list: re-timed, record replaced
            c.linear_act(x, ws[0], None, o_c, c.ACT_NONE)
            a.linear_act(x, ws[0], None, o_a, a.ACT_NONE)                         # a converges if its (larger) list holds c's pick
            assert c.lib.qp_linear_plan_choice(c.ctx, m, n, k, 0, 0, ctypes.byref(idx), ctypes.byref(kind)) == 0 and kind.value == 2
            assert a.lib.qp_linear_plan_choice(a.ctx, m, n, k, 0, 0, ctypes.byref(idx), ctypes.byref(kind)) == 0
            torch.cuda.synchronize()
            if kind.value == 2:
                assert torch.equal(o_c.view(torch.int16), o_a.view(torch.int16)), (m, n, k)
        # the status no longer shares the return value with the index: bad arguments are an error, not "not planned yet"
        assert c.lib.qp_linear_plan_choice(c.ctx, m, n, k, 7, 0, ctypes.byref(idx), ctypes.byref(kind)) == -1 and idx.value == -1
        assert b"act=7" in c.lib.qp_last_error()
    print(f"shapes whose tuned candidate rounds differently from candidate 0: {differ_from_default} of 4")


def test_one_call_path_falls_back_when_plan_selection_fails(monkeypatch):
    """qp_linear_tune reporting "no candidate ran" for ONE projection shape at one row count must not abort generate():
    segments of that size take the per-operator loop (torch.mm for the failed shape, said once on stderr and recorded in
    engine._TUNE_FAILURES); every other segment size keeps the one-call path.  Result within the engine's tolerance of the untouched run."""
    from quickvideo_amd import engine as E
    from quickvideo_amd.native import QuickPrefillError, QuickPrefillOps
    monkeypatch.setenv("QP_TUNE_GEMMS", "0")
    spec_o, w, plan, pos, delta, embeds = make_case(24, 16, 24, 8, 15, 20)                  # groups of 399 / 384 / 384 tokens, tail 20
    cfg = LVUConfig("x", video_group_size=8, top_p=0.5)
    eng_ok, logits_ok = run_gpu(TINY, w, plan, pos, embeds, cfg)
    assert eng_ok._native_state is not None
    real = QuickPrefillOps.linear_tune
    shared_before = {k: dict(v) for k, v in QuickPrefillEngine._SHARED.items()}

    def failing(self, x, weights, bias, out, act=0, alpha=1.0):
        if x.shape[0] == 384 and weights[0].shape[0] == TINY.hidden and weights[0].shape[1] == TINY.intermediate:     # the down projection of the 384-row groups
            raise QuickPrefillError(-3, "qp_linear_tune: no candidate ran")
        return real(self, x, weights, bias, out, act, alpha)
    monkeypatch.setattr(QuickPrefillOps, "linear_tune", failing)
    for d in QuickPrefillEngine._SHARED.values():                   # a fresh process: nothing tuned, nothing planned yet
        d.clear()
    E._TUNE_FAILURES.clear()
    try:
        eng, logits = run_gpu(TINY, w, plan, pos, embeds, cfg)
        assert list(eng.arena.len) == list(eng_ok.arena.len)
        assert any(k[0] == "lt" and k[1] == 384 for k in E._TUNE_FAILURES), E._TUNE_FAILURES
        plans = {k[1]: v for k, v in eng._gemm_plans.items() if k[0] == "native"}
        assert plans[384] == E._NO_NATIVE and plans[plan.tokens[0]] != E._NO_NATIVE and plans[plan.tail_len] != E._NO_NATIVE
        check_logits(logits.numpy(), logits_ok.numpy())
    finally:
        for k, d in QuickPrefillEngine._SHARED.items():            # leave the process-wide tables as the other tests expect them
            d.clear(); d.update(shared_before.get(k, {}))
        E._TUNE_FAILURES.clear()
