"""CPU tests of the engine's host logic with the oracle-backed ops double (tests/oracle_ops.py): group loop,
arena bookkeeping, hidden-state pruning hand-off, prompt tail, decode positions — compared with the oracle's own
end-to-end restatement and with the golden vectors from the reference (GV5)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import qp_oracle as O
from oracle.make_golden import E2E_CASES, TINY as TINY_DICT
from quickvideo_amd import planner
from quickvideo_amd.engine import QuickPrefillEngine
from quickvideo_amd.lvu_config import LVUConfig, LVULayerConfig, effective_k
from quickvideo_amd.spec import TINY
from quickvideo_amd.weights import DecoderWeights
from tests.oracle_ops import OracleOps


def make_case(frames, gh, gw, gs, prefix, tail, seed=4242):
    spec_o = O.TextSpec(**TINY_DICT)
    w = {k: v.to(torch.bfloat16) for k, v in O.synthetic_text_weights(spec_o, seed=7, norm_jitter=0.1).items()}
    n_video = (frames // 2) * (gh // 2) * (gw // 2)
    T = prefix + n_video + tail
    plan = planner.plan_groups(frames, gs, gh, gw, prefix, T)
    pos, delta = planner.mrope_positions(prefix, (frames // 2, gh, gw), tail)
    rs = np.random.RandomState(seed)
    embeds = torch.from_numpy(rs.standard_normal((T, spec_o.hidden)).astype(np.float32) * 0.5).to(torch.bfloat16)
    return spec_o, w, plan, pos, delta, embeds


def run_engine(w, plan, pos, embeds, cfg, ops, tp_rank=0, tp_size=1, tp_group=None):
    dw = DecoderWeights.from_named(TINY, w, "cpu", tp_rank=tp_rank, tp_size=tp_size)
    eng = QuickPrefillEngine(dw, cfg, capacity=embeds.shape[0] + 8, max_group_tokens=max(plan.tokens + [plan.tail_len]), device="cpu",
                             ops=ops, tp_group=tp_group)
    eng.kept_trace = []
    start = 0
    post = torch.from_numpy(pos)
    for n in plan.tokens:
        eng.prefill_group(embeds[start:start + n], post[:, start:start + n])
        start += n
    logits = eng.prefill_tail(embeds[start:], post[:, start:])
    return eng, logits


@pytest.mark.parametrize("top_p,top_k,pps,decay", [(0.5, None, None, None), (None, 5, None, None), (0.5, None, 1, None),
                                                    (0.6, None, 0, "linear"), (None, None, None, None), (0.3, None, None, "exponential")])
def test_engine_equals_oracle_e2e(top_p, top_k, pps, decay):
    spec_o, w, plan, pos, delta, embeds = make_case(12, 8, 4, 4, 9, 6)
    cfg = LVUConfig("x", top_p=top_p, top_k=top_k, prefill_prune_starting_layer=pps, top_k_decay_type=decay,
                    top_k_decay_factor=0.8 if decay == "exponential" else None, video_group_size=4)
    eng, logits = run_engine(w, plan, pos, embeds, cfg, OracleOps())
    ref = O.group_prefill(w, spec_o, embeds, pos, plan.tokens,
                          O.PruneCfg(top_k=top_k, top_p=top_p, top_k_decay_type=decay, top_k_decay_factor=cfg.top_k_decay_factor,
                                     prefill_prune_starting_layer=pps))
    assert eng.arena.len == ref["cache_len"]
    flat_ref = [k for g in ref["kept"] for k in g]
    assert len(flat_ref) == len(eng.kept_trace)
    for (l, got), want in zip(eng.kept_trace, flat_ref):
        assert (got is None) == (want is None)
        if want is not None:
            assert np.array_equal(got.numpy(), want)
    assert torch.equal(logits, ref["logits"])          # same ops in the same order -> identical
    for l in range(spec_o.n_layers):                   # arena content == oracle cache
        assert torch.equal(eng.arena.k(l)[:, :eng.arena.len[l]], ref["cache"].k[l])
        assert torch.equal(eng.arena.v(l)[:, :eng.arena.len[l]], ref["cache"].v[l])


@pytest.mark.parametrize("ci", [i for i, c in enumerate(E2E_CASES) if c[1] == "bfloat16"])
def test_engine_vs_reference_golden(golden_dir, ci):
    """Engine host logic (with oracle math) against the composite REFERENCE run (GV5)."""
    data = np.load(os.path.join(golden_dir, "gv5_e2e.npz"))
    name, dtn, frames, gh, gw, gs, prefix, tail, top_p, top_k = E2E_CASES[ci]
    spec_o, w, plan, pos, delta, embeds = make_case(frames, gh, gw, gs, prefix, tail)
    eng, logits = run_engine(w, plan, pos, embeds, LVUConfig("x", top_p=top_p, top_k=top_k, video_group_size=gs), OracleOps())
    assert eng.arena.len == list(data[f"{name}_cache_len"])
    ref = data[f"{name}_logits"]
    got = logits.numpy()
    assert np.max(np.abs(got - ref)) <= 3e-2
    assert float(np.dot(got, ref) / (np.linalg.norm(got) * np.linalg.norm(ref))) >= 0.999


def test_effective_k_matches_golden_table(golden_dir):
    rows = json.load(open(os.path.join(golden_dir, "gv3_effective_k.json")))
    for q_len, top_k, top_p, decay, factor, layer, L, enable, want in rows:
        cfg = LVUConfig("x", top_k=top_k, top_p=top_p, top_k_decay_type=decay, top_k_decay_factor=factor, enable=enable)
        if isinstance(want, str):
            with pytest.raises(TypeError):
                effective_k(q_len, cfg, layer, L)
        else:
            assert effective_k(q_len, cfg, layer, L) == want


def test_config_surface():
    """Field names/defaults of lvu/lvu_config.py:3-33 stay drop-in."""
    c = LVUConfig("Qwen/Qwen2-VL-7B-Instruct")
    assert (c.top_k_predict_type, c.top_k, c.top_p, c.do_top_k_for_query, c.adaptive_local_attention, c.num_frames, c.enable,
            c.query_based) == ("key_norms_small", None, None, False, True, 32, True, False)
    assert LVUConfig("x", top_k_decay_type="linear").top_k_decay_factor == 0.5
    assert LVUConfig("x", top_k_predict_type="query_attention_weights").query_based is True
    lc = LVULayerConfig(layer_idx=27, total_layers=28, lvu_config=LVUConfig("x", prefill_prune_starting_layer=20))
    assert lc.is_last_layer and lc.prune_for_next_layer
    assert not LVULayerConfig(layer_idx=3, total_layers=28, lvu_config=LVUConfig("x", prefill_prune_starting_layer=20)).prune_for_next_layer
    # intended top_k_starting_layer semantics (reference crashes there: utils.py:253)
    assert effective_k(100, LVUConfig("x", top_p=0.5, top_k_starting_layer=4), 2, 28) is None
    assert effective_k(100, LVUConfig("x", top_p=0.5, top_k_starting_layer=4), 4, 28) == 50


def test_planner_matches_oracle():
    for args in [(64, 16, 40, 72, 15, 15 + 23040 + 30), (8, 3, 4, 6, 5, 36), (12, 8, 4, 4, 2, 29), (8, 0, 4, 6, 5, 40), (7200 // 2 * 2, 16, 28, 40, 15, 15 + 1008000 + 29)]:
        a, b = planner.plan_groups(*args), O.plan_groups(*args)
        assert (a.tokens, a.grid_thw, a.pixel_rows, a.frames, a.past_len_after, a.tail_len) == (b.tokens, b.grid_thw, b.pixel_rows, b.frames, b.past_len_after, b.tail_len)
    with pytest.raises(TypeError):
        planner.plan_groups(8, None, 4, 6, 5, 40)
    with pytest.raises(AssertionError):
        planner.plan_groups(8, 4, 4, 6, 5, 29)          # no prompt tail left
    for p, g, t, sc in [(5, (3, 4, 6), 7, 1.0), (15, (32, 40, 72), 30, 1.0), (3, (6, 4, 4), 2, 2.0)]:
        pa, da = planner.mrope_positions(p, g, t, temporal_scale=sc); pb, db = O.mrope_positions(p, g, t, temporal_scale=sc)
        assert np.array_equal(pa, pb) and da == db
    assert planner.video_frame_size(64, 1080, 1920) == O.video_frame_size(64, 1080, 1920) == (560, 1008)


def test_decode_step_positions_match_tail_prefill():
    """a10: greedy decode positions = original sequence index + rope_delta on all three streams (the reference keeps the
    caller-supplied cache_position, qwen25_lvu.py:445-464).  Decoding the last prompt token must give the same logits as
    prefetching it as part of the tail."""
    spec_o, w, plan, pos, delta, embeds = make_case(8, 4, 6, 4, 5, 7)
    cfg = LVUConfig("x", top_p=0.5, video_group_size=4)
    T = embeds.shape[0]
    post = torch.from_numpy(pos)

    def run(split_last):
        dw = DecoderWeights.from_named(TINY, w, "cpu")
        eng = QuickPrefillEngine(dw, cfg, capacity=T + 8, max_group_tokens=32, device="cpu", ops=OracleOps())
        st = 0
        for n in plan.tokens:
            eng.prefill_group(embeds[st:st + n], post[:, st:st + n]); st += n
        if not split_last:
            return eng.prefill_tail(embeds[st:], post[:, st:])
        eng.prefill_tail(embeds[st:T - 1], post[:, st:T - 1])
        return eng.decode_step(embeds[T - 1], delta)

    a, b = run(False), run(True)
    assert torch.allclose(a, b, atol=2e-2) and int(a.argmax()) == int(b.argmax())


@pytest.mark.parametrize("mode", ["key_norms", "vector_norms", "vector_norms_small"])
def test_engine_equals_oracle_other_norm_modes(mode):
    """The reference's other norm-based predict types (utils.py:117-131): the engine with the oracle-backed ops double equals
    the composite oracle exactly (kept indices, cache contents, logits)."""
    spec_o, w, plan, pos, delta, embeds = make_case(12, 8, 4, 4, 9, 6)
    cfg = LVUConfig("x", top_p=0.5, video_group_size=4, top_k_predict_type=mode)
    eng, logits = run_engine(w, plan, pos, embeds, cfg, OracleOps())
    ref = O.group_prefill(w, spec_o, embeds, pos, plan.tokens, O.PruneCfg(top_p=0.5, top_k_predict_type=mode))
    base = O.group_prefill(w, spec_o, embeds, pos, plan.tokens, O.PruneCfg(top_p=0.5))
    assert eng.arena.len == ref["cache_len"]
    flat_ref = [k for g in ref["kept"] for k in g]
    flat_base = [k for g in base["kept"] for k in g]
    differs = False
    for (l, got), want, b in zip(eng.kept_trace, flat_ref, flat_base):
        assert (got is None) == (want is None)
        if want is not None:
            assert np.array_equal(got.numpy(), want)
            differs |= not np.array_equal(want, b)
    assert differs                                     # the mode really selects other rows than key_norms_small
    assert torch.equal(logits, ref["logits"])


def test_unknown_predict_type_is_a_value_error():
    spec_o, w, plan, pos, delta, embeds = make_case(12, 8, 4, 4, 9, 6)
    with pytest.raises(ValueError, match="Unknown predict type"):          # lvu/utils.py:189
        run_engine(w, plan, pos, embeds, LVUConfig("x", top_p=0.5, top_k_predict_type="salient_tokens"), OracleOps())


@pytest.mark.parametrize("ci", [1, 2])
def test_engine_decode_vs_reference_golden(golden_dir, ci):
    """GV7 (the bf16 cases: the engine computes in bf16): the engine (host logic + oracle-backed ops double) decoding the reference's own greedy tokens reproduces the
    composite reference's per-step logits, next tokens and final cache lengths."""
    from tests.test_oracle_golden import gv7_case
    c = gv7_case(golden_dir, ci)
    dw = DecoderWeights.from_named(TINY, c["w"], "cpu")
    cfg = LVUConfig("x", top_p=c["top_p"], top_k=c["top_k"], video_group_size=c["gs"])
    eng = QuickPrefillEngine(dw, cfg, capacity=c["T"] + 8, max_group_tokens=max(c["plan"].tokens + [c["plan"].tail_len]), device="cpu",
                             ops=OracleOps())
    post, st = torch.from_numpy(c["pos"]), 0
    for n in c["plan"].tokens:
        eng.prefill_group(c["embeds"][st:st + n], post[:, st:st + n]); st += n
    logits = eng.prefill_tail(c["embeds"][st:], post[:, st:])
    tol = 3e-2                                            # stated bf16 tolerance of the GV5 / GV7 oracle tests
    assert np.max(np.abs(logits.numpy() - c["tail_logits"])) <= tol
    tok = int(torch.argmax(logits))
    for i, fed in enumerate(c["tokens"]):
        assert tok == fed
        lg = eng.decode_step(eng.embed_tokens(torch.tensor([fed])), c["delta"])
        assert np.max(np.abs(lg.numpy() - c["decode_logits"][i])) <= tol
        tok = int(torch.argmax(lg))
    assert eng.arena.len == c["cache_len"]


def test_split_gate_up_path_equals_fused(monkeypatch):
    """The engine may run gate and up as two GEMMs (faster in hipBLASLt for some segment sizes): same result as the fused one."""
    spec_o, w, plan, pos, delta, embeds = make_case(8, 4, 6, 4, 5, 7)
    cfg = LVUConfig("x", top_p=0.5, video_group_size=4)
    outs = []
    for rows in ("1000000,1000001", "0,1000000"):
        monkeypatch.setenv("QP_SPLIT_GATE_UP_ROWS", rows)
        eng = QuickPrefillEngine(DecoderWeights.from_named(TINY, w, "cpu"), cfg, capacity=embeds.shape[0] + 8, max_group_tokens=32, device="cpu",
                                 ops=OracleOps())
        post, st = torch.from_numpy(pos), 0
        for n in plan.tokens:
            eng.prefill_group(embeds[st:st + n], post[:, st:st + n]); st += n
        outs.append(eng.prefill_tail(embeds[st:], post[:, st:]))
    assert torch.allclose(outs[0], outs[1], atol=2e-2) and int(outs[0].argmax()) == int(outs[1].argmax())


@pytest.mark.parametrize("pt", ["query_attention_weights", "query_attention_weights_by_value_norm"])
def test_query_based_groups_host_logic(pt):
    """prompt tokens appended to every group, their K/V kept out of the cache, flash-attn's bottom-right alignment for
    n+m queries over past+n keys done as two launches (prefix-only rows + the standard launch shifted by m), k highest scores kept.
    Engine (oracle-backed ops) == the oracle's monolithic restatement: identical kept lists, cache lengths and logits."""
    spec_o, w, plan, pos, delta, embeds = make_case(24, 12, 16, 8, 15, 20)
    m = plan.tail_len
    cfg = LVUConfig("x", top_p=0.5, video_group_size=8, top_k_predict_type=pt)
    assert cfg.query_based
    eng = QuickPrefillEngine(DecoderWeights.from_named(TINY, w, "cpu"), cfg, capacity=embeds.shape[0] + 8 + m, max_group_tokens=max(plan.tokens) + m,
                             device="cpu", ops=OracleOps())
    eng.kept_trace = []
    post, start, tail = torch.from_numpy(pos), 0, embeds[-m:]
    with pytest.raises(ValueError):
        eng.prefill_group(embeds[:plan.tokens[0]], post[:, :plan.tokens[0]])            # the prompt rows are mandatory in this mode
    for n in plan.tokens:
        eng.prefill_group(embeds[start:start + n], post[:, start:start + n + m], prompt_embeds=tail)
        start += n
    logits = eng.prefill_tail(embeds[start:], post[:, start:])
    ref = O.group_prefill(w, spec_o, embeds, pos, plan.tokens, O.PruneCfg(top_p=0.5, top_k_predict_type=pt))
    assert eng.arena.len == ref["cache_len"] and eng.seq_pos == embeds.shape[0]
    flat = [k for g in ref["kept"] for k in g]
    for (l, got), want in zip(eng.kept_trace, flat):
        assert (got is None) == (want is None) and (want is None or np.array_equal(got.numpy(), want))
    assert float((logits - ref["logits"]).abs().max()) == 0.0
    # pruning the prompt tail in this mode: the reference asserts (no scores outside the prompt-appended groups, utils.py:56)
    cfg2 = LVUConfig("x", top_p=0.5, video_group_size=8, top_k_predict_type=pt, do_top_k_for_query=True)
    eng2 = QuickPrefillEngine(DecoderWeights.from_named(TINY, w, "cpu"), cfg2, capacity=64, max_group_tokens=64, device="cpu", ops=OracleOps())
    with pytest.raises(AssertionError):
        eng2.prefill_tail(embeds[:20], post[:, :20])
    # and the mode is not the key-norm mode in disguise
    base = O.group_prefill(w, spec_o, embeds, pos, plan.tokens, O.PruneCfg(top_p=0.5))
    assert any(not np.array_equal(a, b) for a, b in zip(flat, [k for g in base["kept"] for k in g]) if a is not None)


def test_bench_cpu_baseline_check_mode_runs_without_a_gpu():
    """`bench.py --cpu-baseline-check CFG` (the calibration of the bounded-sample CPU estimator; only the oracle runs) prints one JSON
    object with the measured / predicted seconds and their ratio."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-baseline-check", "tiny"], capture_output=True, text=True, timeout=300, cwd=root)
    assert p.returncode == 0, p.stderr[-1000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["measured"]["layers"] == 4 and d["estimator_sample"]["layers"] == 1 and d["estimator_over_measured_speed"] > 0


def test_switch_interval_override_is_ref_counted():
    """Two overlapping generate() calls (two LVU objects in one process): the process-wide thread switch interval is lowered by the first
    to enter and restored by the LAST to leave — never restored under a running call, never left lowered."""
    import sys
    from quickvideo_amd.pipeline import _switch_interval_override
    base = sys.getswitchinterval()
    a, b = _switch_interval_override(2e-5), _switch_interval_override(2e-5)
    a.__enter__()
    assert sys.getswitchinterval() == pytest.approx(2e-5)
    b.__enter__()
    a.__exit__(None, None, None)
    assert sys.getswitchinterval() == pytest.approx(2e-5)          # b is still running
    b.__exit__(None, None, None)
    assert sys.getswitchinterval() == pytest.approx(base)
    with _switch_interval_override(0):                             # QP_SWITCH_INTERVAL_S=0: no override at all
        assert sys.getswitchinterval() == pytest.approx(base)


def test_config_fields_outside_the_scope_warn_once_instead_of_being_silently_ignored():
    """cache_dir / save_video_cache (the reference's on-disk frame cache, qwen25_lvu.py:552-592) are accepted for drop-in construction but
    not implemented: the first config that sets them says so (once per process)."""
    import warnings
    from quickvideo_amd import lvu_config
    lvu_config._IGNORED_WARNED.clear()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        lvu_config.LVUConfig("x", save_video_cache=True)
        lvu_config.LVUConfig("x", cache_dir="/tmp/c")
        lvu_config.LVUConfig("x")
    assert len(w) == 1 and "frame cache is not implemented" in str(w[0].message)


def test_ragged_tp_mlp_shards_are_zero_padded_exactly():
    """weights.padded_inter / pad_mlp_shard (round 6): a tensor-parallel MLP column shard whose width is no multiple of 64 (72B: 29568 / 8 =
    3696) is stored zero-padded to the next multiple of 128 — hipBLASLt's down projection is 24-26 % faster on the aligned K
    (profiles/r6j_gemm_alignment_probe.txt).  The padded shard must be the unpadded one plus zeros, in both constructors, and the ranks'
    partial MLP outputs must still add up to the full model's."""
    import dataclasses
    from quickvideo_amd.weights import padded_inter, pad_mlp_shard
    assert [padded_inter(v) for v in (3696, 7392, 3712, 2368, 18944, 88)] == [3712, 7424, 3712, 2368, 18944, 128]
    spec = dataclasses.replace(TINY, intermediate=176, n_layers=1)
    full = DecoderWeights.synthetic(spec, "cpu", seed=4, dtype=torch.float32)
    I, d = spec.intermediate, spec.hidden
    g = torch.Generator().manual_seed(1)
    x = torch.randn(5, d, generator=g, dtype=torch.float64)
    mlp = lambda gu, dn, li: torch.nn.functional.linear(torch.nn.functional.silu(x @ gu[:li].double().t()) * (x @ gu[li:].double().t()), dn.double())
    want = mlp(full.layers[0].w_gate_up, full.layers[0].w_down, I)
    sd = {"embed_tokens.weight": full.embed, "norm.weight": full.norm, "lm_head.weight": full.lm_head}
    lw = full.layers[0]
    H, KV, D = spec.n_heads, spec.n_kv_heads, spec.head_dim
    sd.update({"layers.0.input_layernorm.weight": lw.ln1, "layers.0.post_attention_layernorm.weight": lw.ln2,
               "layers.0.q_proj.weight": lw.w_qkv[:H * D], "layers.0.k_proj.weight": lw.w_qkv[H * D:(H + KV) * D], "layers.0.v_proj.weight": lw.w_qkv[(H + KV) * D:],
               "layers.0.q_proj.bias": lw.b_qkv[:H * D], "layers.0.k_proj.bias": lw.b_qkv[H * D:(H + KV) * D], "layers.0.v_proj.bias": lw.b_qkv[(H + KV) * D:],
               "layers.0.o_proj.weight": lw.w_o, "layers.0.mlp.gate_proj.weight": lw.w_gate_up[:I], "layers.0.mlp.up_proj.weight": lw.w_gate_up[I:],
               "layers.0.mlp.down_proj.weight": lw.w_down})
    for make in (lambda r: DecoderWeights.synthetic(spec, "cpu", seed=4, dtype=torch.float32, tp_rank=r, tp_size=2),
                 lambda r: DecoderWeights.from_named(spec, sd, "cpu", dtype=torch.float32, tp_rank=r, tp_size=2)):
        got = torch.zeros_like(want)
        for r in range(2):
            w = make(r)
            gu, dn = w.layers[0].w_gate_up, w.layers[0].w_down
            assert w.local_inter == 128 and gu.shape == (256, d) and dn.shape == (d, 128)
            lo = r * 88
            assert torch.equal(gu[:88], lw.w_gate_up[lo:lo + 88]) and torch.equal(gu[128:216], lw.w_gate_up[I + lo:I + lo + 88])
            assert torch.equal(dn[:, :88], lw.w_down[:, lo:lo + 88])
            assert not gu[88:128].any() and not gu[216:].any() and not dn[:, 88:].any()
            got += mlp(gu, dn, 128)
        assert torch.allclose(got, want, rtol=0, atol=1e-12)
    same = pad_mlp_shard(lw.w_gate_up[:256], lw.w_down[:, :128])           # an aligned shard is handed back as it is
    assert same[0].shape[0] == 256 and same[1].shape[1] == 128
