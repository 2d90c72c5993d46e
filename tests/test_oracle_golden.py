"""Pins oracle/qp_oracle.py against the golden vectors produced by the REFERENCE's own functions
(oracle/make_golden.py, run in the build container).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import qp_oracle as O
from oracle.make_golden import COMPACT_CASES, E2E_CASES, SELECT_CASES, TINY, make_keys


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def gv1(golden_dir):
    return np.load(os.path.join(golden_dir, "gv1_select.npz")), json.load(open(os.path.join(golden_dir, "gv1_select.json")))


@pytest.mark.parametrize("ci", range(len(SELECT_CASES)))
def test_select_matches_reference(gv1, ci):
    data, meta = gv1
    m = meta[ci]
    dist, hkv, n, k = SELECT_CASES[ci]
    assert (m["dist"], m["hkv"], m["n"], m["k"]) == (dist, hkv, n, k)
    keys = make_keys(dist, hkv, n, m["seed"])
    bits = O.torch_bf16_to_bits(keys[0])
    norms = O.key_norms_bf16(O.key_sumsq_heads(bits))
    tnorm = data[f"c{ci}_torch_norm_bits"]
    differ = np.nonzero(norms != tnorm)[0]
    # canonical summation order vs torch's: identical bf16 norms except the recorded rounding-boundary rows
    # (<= 1 on the natural distributions; the deliberately adversarial "rounding_boundary" case sits ON the fp32 -> bf16 tie: ~30 % of its rows)
    assert len(differ) == m["norm_rows_differ"] and (len(differ) <= 1 or dist == "rounding_boundary")
    if dist == "rounding_boundary":
        assert len(differ) >= 100
    if len(differ):
        assert np.all(np.abs(norms[differ].astype(int) - tnorm[differ].astype(int)) == 1)
    idx = O.select_k_smallest(tnorm, k)            # select on the reference's own norms: isolates the tie rule
    # (1) bit-exact vs the reference with its deployment (stable) sort
    assert np.array_equal(idx, data[f"c{ci}_ref_idx_stable"])
    # (2) vs the raw CPU reference: exact when the k-th place is not inside a tie class, else threshold property
    ref = data[f"c{ci}_ref_idx"]
    tau = m["tau"]
    if not m["boundary_tied"]:
        assert np.array_equal(idx, ref)
    for got in (idx, ref):
        assert len(got) == k and np.all(np.diff(got) > 0)
        kept = np.zeros(n, bool); kept[got] = True
        assert np.all(kept[tnorm < tau]) and not np.any(kept[tnorm > tau])
    assert len(set(idx) ^ set(ref)) == m["sym_diff_cpu_vs_stable"]
    # (3) full oracle path (own norms) agrees with the reference whenever the norms agree
    own = O.select_k_smallest(norms, k)
    if len(differ) == 0:
        assert np.array_equal(own, data[f"c{ci}_ref_idx_stable"])
    else:
        check_select_where_norms_differ(own, data[f"c{ci}_ref_idx_stable"], norms, tnorm, k)


def check_select_where_norms_differ(own, ref_stable, norms, tnorm, k):
    """What "index parity" still pins when the canonical fp32 sum order (HIP kernel = oracle) rounds some rows' bf16 norm one ulp away from
    torch's (utils.py:134-135 reduces in torch's vectorised order): the two kept lists may differ ONLY on rows whose norm differs, or
    whose norm lies in the band between the two thresholds (tie class members included); every agreeing row strictly below both
    thresholds is kept by both, strictly above both by neither."""
    n = len(norms)
    differ = norms != tnorm
    assert np.all(np.abs(norms[differ].astype(int) - tnorm[differ].astype(int)) == 1)        # one bf16 ulp, never more
    tau_o, tau_t = O.select_threshold(norms, k)[0], O.select_threshold(tnorm, k)[0]
    lo, hi = min(tau_o, tau_t), max(tau_o, tau_t)
    ko = np.zeros(n, bool); ko[own] = True
    kt = np.zeros(n, bool); kt[ref_stable] = True
    agree = ~differ
    assert np.all(ko[agree & (norms < lo)]) and np.all(kt[agree & (norms < lo)])
    assert not np.any(ko[agree & (norms > hi)]) and not np.any(kt[agree & (norms > hi)])
    moved = ko != kt
    in_band = ((norms >= lo) & (norms <= hi)) | ((tnorm >= lo) & (tnorm <= hi))
    assert np.all(differ[moved] | in_band[moved]), np.nonzero(moved & ~differ & ~in_band)[0]
    return int(moved.sum())


def test_effective_k_table(golden_dir):
    rows = json.load(open(os.path.join(golden_dir, "gv3_effective_k.json")))
    assert len(rows) > 10000
    n_prune = 0
    for q_len, top_k, top_p, decay, factor, layer, L, enable, want in rows:
        if isinstance(want, str):      # the reference raises (top_k None * decay): we must raise too
            with pytest.raises(TypeError):
                O.effective_k(q_len, top_k, top_p, decay, factor, layer, L, enable)
            continue
        got = O.effective_k(q_len, top_k, top_p, decay, factor, layer, L, enable)
        assert got == want, (q_len, top_k, top_p, decay, factor, layer, enable, got, want)
        n_prune += want is not None
    assert n_prune > 1000


def test_effective_k_spot_values():
    # SURVEY.md §8c verified outputs of the reference
    ek = lambda q, k=None, p=None, **kw: O.effective_k(q, k, p, kw.get("decay"), kw.get("factor"), kw.get("layer", 0), 28, True)
    assert ek(5775, p=0.5) == 2887 and ek(5760, p=0.5) == 2880 and ek(2880, p=0.25) == 720
    assert ek(960, k=64) == 64 and ek(35, p=0.2) == 7 and ek(100, p=0.29) == 28
    assert ek(100, p=1.0) is None and ek(100) is None and ek(1, p=0.5) is None
    assert [ek(100, k=50, decay="linear", factor=0.5, layer=l) for l in (0, 14, 27)] == [50, 25, 2]


@pytest.mark.parametrize("ci", range(len(COMPACT_CASES)))
def test_compaction_matches_reference(golden_dir, ci):
    meta = json.load(open(os.path.join(golden_dir, "gv2_compaction.json")))[ci]
    past, n, k, hkv = COMPACT_CASES[ci]
    rs = np.random.RandomState(meta["seed"])
    keys = torch.from_numpy(rs.standard_normal((1, hkv, past + n, 128)).astype(np.float32)).to(torch.bfloat16)
    vals = torch.from_numpy(rs.standard_normal((1, hkv, past + n, 128)).astype(np.float32)).to(torch.bfloat16)
    hid = rs.standard_normal((1, n, 16)).astype(np.float32)
    kc, vc = O.torch_bf16_to_bits(keys[0]).copy(), O.torch_bf16_to_bits(vals[0]).copy()
    idx, _ = O.prune_tail(kc, vc, past, n, k)
    assert meta["out_len"] == past + k
    assert sha(kc[:, :past + k]) == meta["k_sha"] and sha(vc[:, :past + k]) == meta["v_sha"]
    # hidden-state pruning hand-off (prune_for_next_layer): same index list gathers hidden / positions
    pos_ids = np.tile(np.arange(n)[None, None], (3, 1, 1)) + 7
    cache_pos = np.arange(n) + past
    pe0 = rs.standard_normal((3, 1, n, 8)).astype(np.float32); pe1 = rs.standard_normal((3, 1, n, 8)).astype(np.float32)
    assert sha(hid[:, idx]) == meta["hidden_sha"] and meta["hidden_shape"] == [1, k, 16]
    assert sha(pos_ids[:, :, idx].astype(np.int64)) == meta["pos_sha"]
    assert sha(cache_pos[idx].astype(np.int64)) == meta["cache_pos_sha"]
    assert sha(pe0[:, :, idx], pe1[:, :, idx]) == meta["pe_sha"]


def test_rope_index_matches_hf(golden_dir):
    for r in json.load(open(os.path.join(golden_dir, "gv6_rope_index.json"))):
        pos, delta = O.mrope_positions(r["prefix"], (r["t"], r["gh"], r["gw"]), r["tail"])
        assert sha(pos.astype(np.int64)) == r["pos_sha"], r
        assert delta == r["delta"] and [int(x) for x in pos[:, -1]] == r["last"]


def test_planner_hand_cases():
    # cfg2: F=64, gs=16, 560x1008 -> grid 40x72, 720 tokens per 2 frames
    p = O.plan_groups(64, 16, 40, 72, 15, 15 + 23040 + 30)
    assert p.tokens == [5775, 5760, 5760, 5760] and p.grid_thw == [(8, 40, 72)] * 4 and p.pixel_rows == [23040] * 4
    assert p.past_len_after == 23055 and p.tail_len == 30
    # odd group size is rounded up to a multiple of temporal_patch_size (qwen25_lvu.py:625-626)
    p = O.plan_groups(8, 3, 4, 6, 5, 5 + 24 + 7)
    assert p.frames == [4, 4] and p.tokens == [17, 12]
    # ragged last group
    p = O.plan_groups(12, 8, 4, 4, 2, 2 + 24 + 3)
    assert p.frames == [8, 4] and p.tokens == [2 + 16, 8] and p.grid_thw == [(4, 4, 4), (2, 4, 4)] and p.pixel_rows == [64, 32]
    # video_group_size == 0 -> single group (timing_baseline.sh:9)
    p = O.plan_groups(8, 0, 4, 6, 5, 40)
    assert p.tokens == [29] and p.tail_len == 11
    # frame size arithmetic (smart_resize + pixel budget) for the bench configs
    assert O.video_frame_size(64, 1080, 1920) == (560, 1008) and O.video_frame_size(256, 1080, 1920) == (280, 504)
    assert O.video_frame_size(512, 1080, 1920) == (224, 420) and O.video_frame_size(7200, 1080, 1920) == (224, 420)


def test_video_plan_vs_reference_golden(golden_dir):
    """GV4 (round 4): frame count and frame size from the VIDEO ENTRY of the message — outputs of the reference's own `smart_nframes` and
    of `fetch_video`'s budget / resize statements (AST-extracted in the build container, oracle/make_golden.py::extract_video_planning).
    Both the oracle's restatement and the PRODUCT's planner must reproduce every row: nframes (or the same exception class and message),
    the clamped max_pixels, the resized (H, W) and whether the reference warned.  Includes the two cases the round-3 review measured
    (max_pixels = 10^7 at 64 frames of 1080x1920 -> 560x1008; max_pixels = 392*560 at 7200 frames -> 252x364)."""
    import logging
    from quickvideo_amd import planner
    d = json.load(open(os.path.join(golden_dir, "gv4_video_plan.json")))
    assert len(d["nframes"]) >= 200 and len(d["frame_size"]) >= 1200
    for r in d["nframes"]:
        for fn in (O.smart_nframes, planner.smart_nframes):
            try:
                got = {"nframes": fn(dict(r["ele"]), r["total_frames"], r["video_fps"])}
            except (ValueError, AssertionError) as e:
                got = {"raises": type(e).__name__, "message": str(e)}
            want = {k: r[k] for k in ("nframes", "raises", "message") if k in r}
            assert got == want, (fn.__module__, r, got)
            if "nframes" in got:
                assert type(got["nframes"]) is type(want["nframes"]) or float(got["nframes"]) == float(want["nframes"])
    records = []
    h = logging.Handler(); h.emit = records.append
    lg = logging.getLogger(planner.__name__); lg.addHandler(h)
    try:
        for r in d["frame_size"]:
            ele = dict(r["ele"])
            assert list(O.video_frame_size(r["nframes"], r["height"], r["width"], ele)) == r["resized"], r
            del records[:]
            assert list(planner.video_frame_size(r["nframes"], r["height"], r["width"], ele)) == r["resized"], r
            assert planner.video_pixel_budget(r["nframes"], ele)[1] == r["max_pixels"], r
            assert (len(records) > 0) == r["warned"], r
    finally:
        lg.removeHandler(h)
    assert planner.video_frame_size(64, 1080, 1920, {"max_pixels": 10 ** 7}) == (560, 1008)
    assert planner.video_frame_size(7200, 392, 560, {"max_pixels": 392 * 560}) == (252, 364)


@pytest.mark.parametrize("ci", range(len(E2E_CASES)))
def test_e2e_composite_oracle(golden_dir, ci):
    """Oracle group-prefill vs transformers-5.15 Qwen2-VL + the reference's post_process_kv_cache."""
    data = np.load(os.path.join(golden_dir, "gv5_e2e.npz"))
    meta = json.load(open(os.path.join(golden_dir, "gv5_e2e.json")))[ci]
    name, dtn, frames, gh, gw, gs, prefix, tail, top_p, top_k = E2E_CASES[ci]
    assert meta["name"] == name
    dtype = getattr(torch, dtn)
    spec = O.TextSpec(**TINY)
    w = {k: v.to(dtype) for k, v in O.synthetic_text_weights(spec, seed=meta["weight_seed"], norm_jitter=0.1).items()}
    n_video = (frames // 2) * (gh // 2) * (gw // 2)
    T = prefix + n_video + tail
    plan = O.plan_groups(frames, gs, gh, gw, prefix, T)
    assert plan.tokens == meta["group_tokens"] and plan.tail_len == meta["tail_len"]
    pos, _ = O.mrope_positions(prefix, (frames // 2, gh, gw), tail)
    rs = np.random.RandomState(meta["embed_seed"])
    embeds = torch.from_numpy(rs.standard_normal((T, spec.hidden)).astype(np.float32) * 0.5).to(dtype)
    out = O.group_prefill(w, spec, embeds, pos, plan.tokens, O.PruneCfg(top_k=top_k, top_p=top_p))
    assert out["cache_len"] == list(data[f"{name}_cache_len"])
    # per-(group,layer) cache-length trace: (layer, before, after) as the reference hook saw it
    trace = [(l, b, a) for (l, b, a) in meta["trace"]]
    ours = []
    run = [0] * spec.n_layers
    for gi, n in enumerate(plan.tokens + [plan.tail_len]):
        for l in range(spec.n_layers):
            kept = out["kept"][gi][l]
            before = run[l] + n
            run[l] = run[l] + (n if kept is None else len(kept))
            ours.append((l, before, run[l]))
    assert ours == trace
    ref = data[f"{name}_logits"]
    got = out["logits"].numpy()
    if dtype == torch.float32:
        assert np.max(np.abs(got - ref)) <= 2e-5, np.max(np.abs(got - ref))
    else:   # bf16: tolerance stated in DESIGN.md (logits |.|max ~1): 3e-2 abs and cosine >= 0.999
        assert np.max(np.abs(got - ref)) <= 3e-2, np.max(np.abs(got - ref))
        cos = float(np.dot(got, ref) / (np.linalg.norm(got) * np.linalg.norm(ref)))
        assert cos >= 0.999, cos


# ---------------------------------------------------------------- GV7: greedy decode after the prompt tail
def gv7_case(golden_dir, ci):
    from oracle.make_golden import E2E_DECODE_CASES
    data = np.load(os.path.join(golden_dir, "gv7_e2e_decode.npz"))
    meta = json.load(open(os.path.join(golden_dir, "gv7_e2e_decode.json")))[ci]
    name, dtn, frames, gh, gw, gs, prefix, tail, top_p, top_k = E2E_DECODE_CASES[ci]
    assert meta["name"] == name
    dtype = getattr(torch, dtn)
    spec = O.TextSpec(**TINY)
    w = {k: v.to(dtype) for k, v in O.synthetic_text_weights(spec, seed=meta["weight_seed"], norm_jitter=0.1).items()}
    w["embed_tokens.weight"] = (w["embed_tokens.weight"] * meta["decode_embed_scale"]).to(dtype)     # make_golden.e2e_case
    n_video = (frames // 2) * (gh // 2) * (gw // 2)
    T = prefix + n_video + tail
    plan = O.plan_groups(frames, gs, gh, gw, prefix, T)
    pos, delta = O.mrope_positions(prefix, (frames // 2, gh, gw), tail)
    assert delta == meta["rope_delta"]
    rs = np.random.RandomState(meta["embed_seed"])
    embeds = torch.from_numpy(rs.standard_normal((T, spec.hidden)).astype(np.float32) * 0.5).to(dtype)
    return dict(name=name, dtype=dtype, spec=spec, w=w, plan=plan, pos=pos, delta=delta, embeds=embeds, T=T, top_p=top_p, top_k=top_k, gs=gs,
                tokens=meta["decode_tokens"], tail_logits=data[f"{name}_logits"], decode_logits=data[f"{name}_decode_logits"],
                cache_len=list(data[f"{name}_cache_len"]))


@pytest.mark.parametrize("ci", range(3))
def test_e2e_decode_composite_oracle(golden_dir, ci):
    """a10: the reference run (HF Qwen2-VL + the reference's prune hook) continued by greedy decode steps over the pruned cache.
    The oracle states decode as PREFILL of the sequence extended by the fed tokens (no pruning for tail segments): its
    last-position logits must reproduce every decode step's logits and greedy token."""
    c = gv7_case(golden_dir, ci)
    steps = len(c["tokens"])
    assert int(np.argmax(c["tail_logits"])) == c["tokens"][0]
    for i in range(steps):
        ext = torch.cat([c["embeds"], c["w"]["embed_tokens.weight"][torch.tensor(c["tokens"][:i + 1])]], 0)
        pos_ext = np.concatenate([c["pos"], np.tile(np.arange(c["T"] + c["delta"], c["T"] + c["delta"] + i + 1, dtype=np.int64), (3, 1))], 1)
        out = O.group_prefill(c["w"], c["spec"], ext, pos_ext, c["plan"].tokens, O.PruneCfg(top_k=c["top_k"], top_p=c["top_p"]))
        got, ref = out["logits"].numpy(), c["decode_logits"][i]
        if c["dtype"] == torch.float32:
            assert np.max(np.abs(got - ref)) <= 2e-5, np.max(np.abs(got - ref))
        else:
            assert np.max(np.abs(got - ref)) <= 3e-2, np.max(np.abs(got - ref))
            assert float(np.dot(got, ref) / (np.linalg.norm(got) * np.linalg.norm(ref))) >= 0.999
        if i + 1 < steps:
            assert int(np.argmax(got)) == c["tokens"][i + 1]
        if i == steps - 1:
            assert out["cache_len"] == c["cache_len"]


# ---------------------------------------------------------------- GV1b: the other norm-based predict types
@pytest.mark.parametrize("mode", ["key_norms", "vector_norms", "vector_norms_small"])
def test_select_other_norm_modes_match_reference(golden_dir, gv1, mode):
    """key_norms / vector_norms(_small) (utils.py:117-131): same norms, other end of the order and/or the value rows.
    Indices from the reference import (stable argsort) must equal the oracle's select on the reference's own norm patterns,
    and on the oracle's norms wherever the two norm computations agree (GV1 records the rare 1-ulp rows)."""
    from oracle.make_golden import MODE_CASES
    data = np.load(os.path.join(golden_dir, "gv1b_select_modes.npz"))
    _, meta1 = gv1
    source, order = O.NORM_PRUNE_MODES[mode]
    sel = O.select_k_largest if order else O.select_k_smallest
    for ci in MODE_CASES:
        dist, hkv, n, k = SELECT_CASES[ci]
        x = make_keys(dist, hkv, n, 1000 + ci)           # the scored rows (keys or values: same recipe)
        tnorm = O.torch_bf16_to_bits(x[0].transpose(0, 1).flatten(1, 2).norm(2, dim=-1))
        ref = data[f"c{ci}_{mode}"]
        assert np.array_equal(sel(tnorm, k), ref), (mode, ci)
        if meta1[ci]["norm_rows_differ"] == 0:
            norms = O.key_norms_bf16(O.key_sumsq_heads(O.torch_bf16_to_bits(x[0])))
            assert np.array_equal(sel(norms, k), ref), (mode, ci)


@pytest.mark.parametrize("ci", range(4))
def test_e2e_composite_oracle_other_norm_modes(golden_dir, ci):
    """GV5b: the oracle's key_norms / vector_norms(_small) paths end to end against transformers-5.15 Qwen2-VL + the reference's
    post_process_kv_cache run with that top_k_predict_type (fp32: logits to 2e-5, i.e. the same rows were kept at every layer)."""
    from oracle.make_golden import E2E_MODE_CASES
    data = np.load(os.path.join(golden_dir, "gv5b_e2e_modes.npz"))
    meta = json.load(open(os.path.join(golden_dir, "gv5b_e2e_modes.json")))[ci]
    name, dtn, frames, gh, gw, gs, prefix, tail, top_p, top_k, mode = E2E_MODE_CASES[ci]
    assert meta["name"] == name and meta["predict_type"] == mode
    dtype = getattr(torch, dtn)
    spec = O.TextSpec(**TINY)
    w = {k: v.to(dtype) for k, v in O.synthetic_text_weights(spec, seed=meta["weight_seed"], norm_jitter=0.1).items()}
    T = prefix + (frames // 2) * (gh // 2) * (gw // 2) + tail
    plan = O.plan_groups(frames, gs, gh, gw, prefix, T)
    pos, _ = O.mrope_positions(prefix, (frames // 2, gh, gw), tail)
    rs = np.random.RandomState(meta["embed_seed"])
    embeds = torch.from_numpy(rs.standard_normal((T, spec.hidden)).astype(np.float32) * 0.5).to(dtype)
    out = O.group_prefill(w, spec, embeds, pos, plan.tokens, O.PruneCfg(top_k=top_k, top_p=top_p, top_k_predict_type=mode))
    assert out["cache_len"] == list(data[f"{name}_cache_len"])
    got, ref = out["logits"].numpy(), data[f"{name}_logits"]
    if dtype == torch.float32:
        assert np.max(np.abs(got - ref)) <= 2e-5, np.max(np.abs(got - ref))
    else:
        assert np.max(np.abs(got - ref)) <= 3e-2, np.max(np.abs(got - ref))
        assert float(np.dot(got, ref) / (np.linalg.norm(got) * np.linalg.norm(ref))) >= 0.999
    base = O.group_prefill(w, spec, embeds, pos, plan.tokens, O.PruneCfg(top_k=top_k, top_p=top_p))
    assert np.max(np.abs(base["logits"].numpy() - ref)) > 1e-3          # the mode matters: default scoring gives other logits


def test_deep_fixture_first_layers_match_oracle(golden_dir):
    """GV8 (28 layers at the 7B dims, made by the reference composite): the oracle re-runs layers 0-1 of group 0 on the CPU from
    the same hash-generated weights / input rows and must reproduce the reference's kept index lists (layer 0: identical inputs,
    so the lists agree except where a bf16 key norm sits on a rounding boundary between the two GEMM implementations)."""
    import json
    meta = json.load(open(os.path.join(golden_dir, "gv8_deep.json")))
    gold = np.load(os.path.join(golden_dir, "gv8_deep.npz"))
    s = dict(meta["spec"]); s["n_layers"] = 2
    spec = O.TextSpec(**s)
    w = O.hashed_text_weights(spec, seed=meta["weight_seed"], with_embed=False)
    n0 = meta["group_tokens"][0]
    T = meta["prefix"] + (meta["frames"] // 2) * (meta["grid_h"] // 2) * (meta["grid_w"] // 2) + meta["tail"]
    emb = O.hashed_normal((T, spec.hidden), meta["embed_seed"], 0.5)[:n0]
    pos, _ = O.mrope_positions(meta["prefix"], (meta["frames"] // 2, meta["grid_h"], meta["grid_w"]), meta["tail"])
    cache = O.OracleCache(2)
    cos, sin = O.mrope_cos_sin(torch.from_numpy(pos[:, :n0]), spec, torch.bfloat16)
    h = emb
    k_keep = O.effective_k(n0, None, meta["top_p"], None, None, 0, meta["spec"]["n_layers"])
    assert k_keep == int(gold["kept_g0_l0"].shape[0])
    with torch.no_grad():
        for l in range(2):
            h, kept, cos, sin = O.decoder_layer(h, w, l, spec, cache, cos, sin, k_keep)
            want = gold[f"kept_g0_l{l}"].astype(np.int64)
            ov = len(set(kept.tolist()) & set(want.tolist())) / len(want)
            assert ov >= (0.99 if l == 0 else 0.95), (l, ov)


def test_query_scores_match_reference(golden_dir):
    """GV9: LVUCache.update's query-based scores (lvu_cache.py:97-117) and the kept lists of both query predict types
    (utils.py:55-62, argsort forced stable) — the oracle restatement reproduces the reference's own outputs bit for bit."""
    import json
    from oracle.make_golden import QUERY_CASES, make_query_case
    data = np.load(os.path.join(golden_dir, "gv9_query_scores.npz"))
    for ci, (hq, hkv, n, m, k) in enumerate(QUERY_CASES):
        q, kk, vv = make_query_case(ci)
        sc = O.query_attention_scores(q[0, :, n:], kk[0, :, :n].contiguous())
        assert np.array_equal(O.torch_bf16_to_bits(sc), data[f"c{ci}_score_bits"])
        assert np.array_equal(O.select_k_largest(O.query_score_keys(sc), k), data[f"c{ci}_query_attention_weights"])
        assert np.array_equal(O.select_k_largest(O.query_score_keys(sc, vv[0, :, :n]), k), data[f"c{ci}_query_attention_weights_by_value_norm"])


def test_full_depth_calibration_record_is_consistent(golden_dir):
    """tests/golden/gv8_deep_oracle_calibration.json (oracle/calibrate_deep.py: the CPU oracle's FULL 28-layer run against GV8, 13 min of
    CPU, not repeated here) is what tests/test_gpu_deep.py derives its bars from.  It must belong to the committed fixture (same argmax,
    cache lengths equal, one overlap figure per layer, layer 0 exact) and say what DESIGN.md quotes."""
    import json
    meta = json.load(open(os.path.join(golden_dir, "gv8_deep.json")))
    cal = json.load(open(os.path.join(golden_dir, "gv8_deep_oracle_calibration.json")))
    assert cal["cache_len_equal"] and cal["argmax"] == cal["reference_argmax"] == meta["argmax"]
    ov = cal["overlap_min_over_groups_by_layer"]
    assert len(ov) == meta["spec"]["n_layers"] and len(cal["overlap_by_group_layer"]) == len(meta["group_tokens"])
    assert ov[0] == 1.0 and min(ov) >= 0.96
    assert abs(cal["logits_max_abs_diff"] - 0.2129) < 1e-3 and abs(cal["logits_cosine"] - 0.99894) < 1e-5
    assert abs(cal["logit_absmax"] - meta["logit_absmax"]) < 1e-6


def test_costed_frame_source_is_deterministic_and_threaded():
    """synthetic://...&decode_h=&decode_w= (frames.py): a frame source that costs what a decoder costs — every frame produced at the
    decode size and LANCZOS-resized on the reader's worker threads.  Same pixels whatever the thread count or access order; the
    calibrated per-frame pad (decode_s) stretches a group to at least its share of the budget."""
    import time
    from quickvideo_amd.frames import open_video
    url = "synthetic://?frames=64&h=112&w=168&fps=2&seed=3&decode_h=270&decode_w=480"
    got = []
    for nt in (1, 4):
        r = open_video(url, num_threads=nt)
        r.height, r.width, r.frame_iter = 112, 168, 8
        r.process(np.arange(0, 32, 2))
        got.append(torch.cat([next(r), next(r)]))
        assert got[-1].shape == (16, 3, 112, 168) and got[-1].dtype == torch.uint8
        with pytest.raises(StopIteration):
            next(r)
    assert torch.equal(got[0], got[1])
    r = open_video(url, num_threads=2)
    r.height, r.width, r.frame_iter = 112, 168, 4
    r.process(np.array([30, 2]))                         # other order, other grouping: frame 2 is still frame 2
    assert torch.equal(next(r)[1], got[0][1])
    r = open_video(url + "&decode_s=0.4", num_threads=4)
    r.height, r.width, r.frame_iter = 112, 168, 8
    r.process(np.arange(16))                             # budget: 0.4 s for 16 frames on 4 threads = 0.1 s per frame per thread, 0.2 s per group of 8
    t0 = time.perf_counter(); next(r); dt = time.perf_counter() - t0
    assert 0.18 <= dt < 1.0, dt
