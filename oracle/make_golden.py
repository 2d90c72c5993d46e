"""Generate tests/golden/* from the REFERENCE's own functions (run in the build container only).

  python oracle/make_golden.py            # needs /root/reference (read-only) + transformers

Nothing here travels to the GPU box except its outputs (small .npz/.json fixtures = data: seeded
input recipes and the reference's outputs).  The reference package cannot be imported as a whole
(qwen_vl_utils / deepcodec / flash_attn are absent — SURVEY.md §8c), so the hot-path modules
lvu/lvu_config.py, lvu/lvu_cache.py, lvu/utils.py are loaded individually through a 5-line shim,
and the end-to-end vector comes from the *composite oracle*: installed transformers' Qwen2-VL text
model (random weights, CPU) whose attention modules call the reference's post_process_kv_cache from
a forward hook — the same point the reference calls it (qwen25_lvu.py:183-192).

Two index columns are stored for every select case:
  ref_idx        — raw output of the reference function on torch-CPU (argsort stable=False = introsort)
  ref_idx_stable — the same reference function with torch.Tensor.argsort forced stable, which is how
                   the reference behaves on its deployment device (torch CUDA sort = stable radix sort).
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import sys
import types
import typing

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("QP_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")   # QP_GOLDEN_OUT: regenerate elsewhere and diff
sys.path.insert(0, os.path.dirname(HERE))

from oracle import qp_oracle as O  # noqa: E402


def load_reference():
    import transformers.cache_utils as cu
    for n in ("List", "Dict", "Optional", "Any", "Tuple"):
        if not hasattr(cu, n):
            setattr(cu, n, getattr(typing, n))
    pkg = types.ModuleType("lvu"); pkg.__path__ = [os.path.join(REF, "lvu")]
    sys.modules["lvu"] = pkg
    mods = {}
    for name in ("lvu_config", "lvu_cache", "utils"):
        spec = importlib.util.spec_from_file_location(f"lvu.{name}", os.path.join(REF, "lvu", f"{name}.py"))
        m = importlib.util.module_from_spec(spec); sys.modules[f"lvu.{name}"] = m
        spec.loader.exec_module(m); mods[name] = m
    return mods


class force_stable_argsort:
    """Emulates the reference's CUDA deployment where Tensor.argsort is a stable radix sort."""

    def __enter__(self):
        self.orig = torch.Tensor.argsort
        orig = self.orig

        def stable(t, *a, **kw):
            kw["stable"] = True
            return orig(t, *a, **kw)
        torch.Tensor.argsort = stable

    def __exit__(self, *exc):
        torch.Tensor.argsort = self.orig


# ---------------------------------------------------------------- seeded input recipes (shared with tests)

def make_keys(dist: str, hkv: int, n: int, seed: int, d: int = 128) -> torch.Tensor:
    """bf16 [1, Hkv, n, D] new-key block.  Recipes documented in tests/golden/README.md."""
    rs = np.random.RandomState(seed)
    if dist == "normal":
        x = rs.standard_normal((hkv, n, d))
    elif dist == "heavy":          # per-channel scales with a few RoPE-style outlier channels + per-token scale
        ch = np.exp(rs.standard_normal((hkv, 1, d)) * 0.7)
        ch[:, :, rs.randint(0, d, size=3)] *= 12.0
        tok = np.exp(rs.standard_normal((1, n, 1)) * 0.5)
        x = rs.standard_normal((hkv, n, d)) * ch * tok
    elif dist == "all_equal":      # every token identical -> one tie class
        x = np.tile(rs.standard_normal((hkv, 1, d)), (1, n, 1))
    elif dist == "distinct":       # token t = e0 * (1 + t/256) scaled so every bf16 norm is distinct
        assert n <= 120
        x = np.zeros((hkv, n, d))
        x[0, :, 0] = 1.0 + rs.permutation(n) / 128.0
    elif dist == "few_levels":     # norms drawn from 5 exact levels -> massive ties
        x = np.zeros((hkv, n, d))
        x[0, :, 0] = rs.randint(1, 6, size=n).astype(np.float64)
    elif dist == "rounding_boundary":
        # Every row's EXACT sum of squares is built to equal T^2 (to ~2^-32 relative), T = a bf16 value + half a bf16 ulp: the fp32 norm
        # lands on a bf16 round-to-even tie or one fp32 ulp beside it, so two fp32 summation orders that differ in the last bit of the sum
        # (torch's vectorised reduce vs the canonical per-head order of the HIP kernel / oracle) round the bf16 norm differently on ~30 %
        # of the rows.  Small noise everywhere + five "digits" a_j = bf16(sqrt(remaining)) at random positions.
        def bf16r(v):
            return torch.from_numpy(np.asarray(v, dtype=np.float32)).to(torch.bfloat16).to(torch.float64).numpy()
        x = bf16r(rs.standard_normal((hkv, n, d)) * 0.05)
        b = bf16r(rs.uniform(2, 16, size=n))
        T = b + 2.0 ** (np.floor(np.log2(b)) - 7) / 2
        for _ in range(5):
            h, c = rs.randint(0, hkv, size=n), rs.randint(0, d, size=n)
            x[h, np.arange(n), c] = 0.0
            R = np.maximum(T * T - (x ** 2).sum(axis=(0, 2)), 0)
            a = bf16r(np.sqrt(R))
            a = np.where(a * a > R, bf16r(a * (1 - 2.0 ** -8)), a)
            x[h, np.arange(n), c] = a
    else:
        raise ValueError(dist)
    return torch.from_numpy(x.astype(np.float32)).to(torch.bfloat16)[None]


def sha(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


SELECT_CASES = [  # (dist, Hkv, n, k)
    ("normal", 4, 64, 32), ("normal", 2, 720, 180), ("normal", 4, 2880, 720), ("normal", 4, 5775, 2887),
    ("normal", 8, 960, 480), ("heavy", 4, 5760, 2880), ("heavy", 2, 2880, 1440), ("heavy", 8, 720, 360),
    ("heavy", 4, 2240, 1120), ("all_equal", 4, 257, 100), ("distinct", 4, 100, 37), ("distinct", 2, 64, 63),
    ("few_levels", 4, 1000, 333), ("few_levels", 4, 2880, 720), ("normal", 4, 35, 7), ("normal", 4, 100, 28),
    ("normal", 4, 2, 1), ("heavy", 4, 4097, 1),
    # built ON the fp32 -> bf16 rounding boundary: ~30 % of the rows get a different bf16 norm from torch's reduce order
    # than from the canonical one; exercises the "norms differ" branch of the raw-reference check deliberately
    ("rounding_boundary", 4, 512, 256),
]


def gen_select(ref):
    U = ref["utils"]
    out = {}
    meta = []
    for ci, (dist, hkv, n, k) in enumerate(SELECT_CASES):
        seed = 1000 + ci
        keys = make_keys(dist, hkv, n, seed)
        vals = torch.zeros_like(keys)
        hid = torch.zeros(1, n, 8)
        mask = U.get_top_k_mask_to_predict(None, keys, vals, hid, top_k=k, predict_type="key_norms_small")
        ref_idx = torch.nonzero(mask[0], as_tuple=True)[0].numpy().astype(np.int32)
        with force_stable_argsort():
            mask_s = U.get_top_k_mask_to_predict(None, keys, vals, hid, top_k=k, predict_type="key_norms_small")
        ref_idx_s = torch.nonzero(mask_s[0], as_tuple=True)[0].numpy().astype(np.int32)
        tnorm = keys[0].transpose(0, 1).flatten(1, 2).norm(2, dim=-1)          # utils.py:134-135 verbatim semantics
        tbits = O.torch_bf16_to_bits(tnorm)
        obits = O.key_norms_bf16(O.key_sumsq_heads(O.torch_bf16_to_bits(keys[0])))
        tau, n_lt, n_eq = O.select_threshold(tbits, k)
        out[f"c{ci}_ref_idx"] = ref_idx
        out[f"c{ci}_ref_idx_stable"] = ref_idx_s
        out[f"c{ci}_torch_norm_bits"] = tbits
        meta.append(dict(case=ci, dist=dist, hkv=hkv, n=n, k=k, seed=seed, tau=tau, n_less=n_lt, n_equal=n_eq,
                         boundary_tied=bool(n_lt + n_eq != k), norm_rows_differ=int((tbits != obits).sum()),
                         sym_diff_cpu_vs_stable=int(len(set(ref_idx) ^ set(ref_idx_s)))))
        print(meta[-1])
    np.savez_compressed(os.path.join(OUT, "gv1_select.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "gv1_select.json"), "w"), indent=1)


MODE_CASES = [0, 1, 3, 5, 9, 10, 12, 14, 16]      # SELECT_CASES indices re-used for the other norm-based predict types


def gen_select_modes(ref):
    """GV1b: kept indices of the reference for predict_type in {key_norms, vector_norms, vector_norms_small}
    (utils.py:117-131) — argsort forced stable (ties -> lowest index), same input recipe as GV1.  For the vector_* modes the
    seeded tensor is passed as `values` (and zeros as keys)."""
    U = ref["utils"]
    out, meta = {}, []
    for ci in MODE_CASES:
        dist, hkv, n, k = SELECT_CASES[ci]
        x = make_keys(dist, hkv, n, 1000 + ci)
        zeros = torch.zeros_like(x)
        hid = torch.zeros(1, n, 8)
        for mode in ("key_norms", "vector_norms", "vector_norms_small"):
            keys, vals = (x, zeros) if mode == "key_norms" else (zeros, x)
            with force_stable_argsort():
                mask = U.get_top_k_mask_to_predict(None, keys, vals, hid, top_k=k, predict_type=mode)
            idx = torch.nonzero(mask[0], as_tuple=True)[0].numpy().astype(np.int32)
            assert len(idx) == k
            out[f"c{ci}_{mode}"] = idx
        meta.append(dict(case=ci, dist=dist, hkv=hkv, n=n, k=k, seed=1000 + ci))
    np.savez_compressed(os.path.join(OUT, "gv1b_select_modes.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "gv1b_select_modes.json"), "w"), indent=1)
    print("gv1b:", len(out), "index lists")


def gen_effective_k(ref):
    U, C = ref["utils"], ref["lvu_config"]
    rows = []
    L = 28
    for q_len in (1, 2, 15, 35, 100, 960, 2240, 2880, 5760, 5775):
        for top_k in (None, 1, 64, 5760):
            for top_p in (None, 0.0, 0.1, 0.2, 0.25, 0.29, 0.5, 0.75, 1.0):
                for decay, factor in ((None, None), ("linear", 0.5), ("exponential", 0.9), ("exponential", 0.33)):
                    for layer in (0, 1, 14, 27):
                        for enable in (True, False):
                            cfg = C.LVUConfig(model_name_or_path="x", top_k=top_k, top_p=top_p, top_k_decay_type=decay,
                                              top_k_decay_factor=factor, enable=enable)
                            lc = C.LVULayerConfig(layer_idx=layer, total_layers=L, lvu_config=cfg)
                            hkv, d = 1, 8
                            keys = torch.arange(q_len, dtype=torch.float32).add(1).view(1, 1, q_len, 1).repeat(1, hkv, 1, d).to(torch.bfloat16)
                            hid = torch.zeros(1, q_len, 4)
                            try:
                                res = U.post_process_kv_cache(hid, None, None, None, None, None, (keys, keys.clone()), lc)
                                kept = res[5][0].shape[2]
                                eff = None if kept == q_len else kept
                            except Exception as ex:   # e.g. top_k None with decay -> TypeError in the reference
                                eff = f"raise:{type(ex).__name__}"
                            rows.append([q_len, top_k, top_p, decay, factor, layer, L, enable, eff])
    json.dump(rows, open(os.path.join(OUT, "gv3_effective_k.json"), "w"))
    print("effective_k rows", len(rows))


COMPACT_CASES = [(0, 15, 7, 4), (2887, 5760, 2880, 4), (8640, 2880, 720, 4), (100, 960, 480, 8), (5, 64, 63, 2), (0, 2240, 1120, 4)]


def gen_compaction(ref):
    U, C = ref["utils"], ref["lvu_config"]
    meta = []
    for ci, (past, n, k, hkv) in enumerate(COMPACT_CASES):
        rs = np.random.RandomState(2000 + ci)
        keys = torch.from_numpy(rs.standard_normal((1, hkv, past + n, 128)).astype(np.float32)).to(torch.bfloat16)
        vals = torch.from_numpy(rs.standard_normal((1, hkv, past + n, 128)).astype(np.float32)).to(torch.bfloat16)
        hid = torch.from_numpy(rs.standard_normal((1, n, 16)).astype(np.float32))
        cfg = C.LVUConfig(model_name_or_path="x", top_k=k, prefill_prune_starting_layer=0)
        lc = C.LVULayerConfig(layer_idx=0, total_layers=4, lvu_config=cfg)
        pos_ids = torch.arange(n)[None, None].repeat(3, 1, 1) + 7
        cache_pos = torch.arange(n) + past
        pe = (torch.from_numpy(rs.standard_normal((3, 1, n, 8)).astype(np.float32)), torch.from_numpy(rs.standard_normal((3, 1, n, 8)).astype(np.float32)))
        with force_stable_argsort():
            h2, am2, pi2, cp2, pe2, (k2, v2) = U.post_process_kv_cache(hid, None, pos_ids, cache_pos, pe, None, (keys, vals), lc)
        meta.append(dict(case=ci, past=past, n=n, k=k, hkv=hkv, seed=2000 + ci,
                         k_sha=sha(O.torch_bf16_to_bits(k2[0])), v_sha=sha(O.torch_bf16_to_bits(v2[0])),
                         out_len=int(k2.shape[2]), hidden_sha=sha(h2.numpy()), pos_sha=sha(pi2.numpy()),
                         cache_pos_sha=sha(cp2.numpy()), pe_sha=sha(pe2[0].numpy(), pe2[1].numpy()),
                         hidden_shape=list(h2.shape), pos_shape=list(pi2.shape)))
        print(meta[-1]["case"], meta[-1]["out_len"], meta[-1]["hidden_shape"])
    json.dump(meta, open(os.path.join(OUT, "gv2_compaction.json"), "w"), indent=1)


# ---------------------------------------------------------------- composite end-to-end oracle (GV5) + rope index (GV6)

TINY = dict(hidden=256, n_heads=2, n_kv_heads=1, head_dim=128, intermediate=512, n_layers=3, vocab=320)

E2E_CASES = [
    # name, dtype, frames, grid_h, grid_w, group_size, prefix, tail, top_p, top_k
    ("fp32_rho1", "float32", 8, 4, 6, 4, 5, 7, None, None),
    ("fp32_rho05", "float32", 8, 4, 6, 4, 5, 7, 0.5, None),
    ("fp32_rho025_g2", "float32", 12, 8, 4, 2, 15, 9, 0.25, None),
    ("fp32_topk5", "float32", 8, 4, 6, 4, 3, 4, None, 5),
    ("bf16_rho05", "bfloat16", 8, 4, 6, 4, 5, 7, 0.5, None),
    ("bf16_rho025_g2", "bfloat16", 12, 8, 4, 2, 15, 9, 0.25, None),
]


def build_hf_tiny(dtype):
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration
    t = TINY
    cfg = Qwen2VLConfig(
        text_config=dict(hidden_size=t["hidden"], num_attention_heads=t["n_heads"], num_key_value_heads=t["n_kv_heads"],
                         intermediate_size=t["intermediate"], num_hidden_layers=t["n_layers"], vocab_size=t["vocab"],
                         rms_norm_eps=1e-6, max_position_embeddings=32768, tie_word_embeddings=False,
                         rope_parameters=dict(rope_type="default", mrope_section=[16, 24, 24], rope_theta=1_000_000.0)),
        vision_config=dict(depth=1, embed_dim=32, hidden_size=t["hidden"], num_heads=2, mlp_ratio=2, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2),
    )
    cfg._attn_implementation = "eager"
    torch.manual_seed(0)
    model = Qwen2VLForConditionalGeneration(cfg).eval()
    spec = O.TextSpec(**t)
    w = O.synthetic_text_weights(spec, seed=7, dtype=torch.float32, norm_jitter=0.1)
    lm = model.model.language_model
    sd = {}
    for k, v in w.items():
        if k == "lm_head.weight":
            continue
        sd[k.replace("layers.", "layers.").replace("q_proj", "self_attn.q_proj").replace("k_proj", "self_attn.k_proj")
           .replace("v_proj", "self_attn.v_proj").replace("o_proj", "self_attn.o_proj")] = v
    missing, unexpected = lm.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in m for m in missing), (missing, unexpected)
    model.lm_head.weight.data.copy_(w["lm_head.weight"])
    return model.to(dtype), spec, w


def e2e_case(ref, name, dtn, frames, gh, gw, gs, prefix, tail, top_p, top_k, predict_type="key_norms_small", decode_steps=0):
    """One composite-oracle run: installed HF Qwen2-VL text stack on CPU + the REFERENCE's post_process_kv_cache hooked after every
    attention.  Returns (final-position logits, per-layer cache lengths, meta).
    decode_steps > 0: continues with that many greedy decode steps over the pruned cache (qwen25_lvu.py:744-761: pruning stays
    off, positions = sequence index + rope_delta on all three streams) and returns (logits, cache_len, meta, decode_logits
    [steps, V]); meta["decode_tokens"] = the token FED at each step (first = argmax of the tail logits)."""
    from transformers import DynamicCache
    U, C = ref["utils"], ref["lvu_config"]
    dtype = getattr(torch, dtn)
    model, spec, w = build_hf_tiny(dtype)
    lm = model.model.language_model
    n_video = (frames // 2) * (gh // 2) * (gw // 2)
    T = prefix + n_video + tail
    plan = O.plan_groups(frames, gs, gh, gw, prefix, T)
    pos, delta = O.mrope_positions(prefix, (frames // 2, gh, gw), tail)
    rs = np.random.RandomState(4242)
    embeds = torch.from_numpy(rs.standard_normal((T, spec.hidden)).astype(np.float32) * 0.5).to(dtype)
    cfg = C.LVUConfig(model_name_or_path="x", top_k=top_k, top_p=top_p, top_k_predict_type=predict_type)
    layer_cfgs = [C.LVULayerConfig(layer_idx=i, total_layers=spec.n_layers, lvu_config=cfg) for i in range(spec.n_layers)]
    cache = DynamicCache(config=model.config.get_text_config())
    kept_trace = []

    def make_hook(i):
        def hook(mod, args, kwargs, output):
            hs = kwargs["hidden_states"]
            lay = cache.layers[i]
            before = lay.keys.shape[2]
            with force_stable_argsort():
                res = U.post_process_kv_cache(hs, None, None, None, None, None, (lay.keys, lay.values), layer_cfgs[i])
            lay.keys, lay.values = res[5]
            kept_trace.append((i, before, lay.keys.shape[2]))
            return output
        return hook
    hooks = [l.self_attn.register_forward_hook(make_hook(i), with_kwargs=True) for i, l in enumerate(lm.layers)]
    start = 0
    segs = plan.tokens + [plan.tail_len]
    post = torch.from_numpy(pos)
    with torch.no_grad():
        for gi, n in enumerate(segs):
            if gi == len(segs) - 1:
                cfg.enable = False        # qwen25_lvu.py:737-738: no pruning for the prompt tail
            pid = post[:, None, start:start + n]
            past_seen = start
            o = lm(inputs_embeds=embeds[None, start:start + n], position_ids=pid, past_key_values=cache, use_cache=True,
                   cache_position=torch.arange(n) + past_seen)
            start += n
        logits = model.lm_head(o.last_hidden_state[:, -1]).float()[0]
        dec_tokens, dec_logits = [], []
        tok = int(torch.argmax(logits))
        if decode_steps:
            # the synthetic embedding table has std 0.02 while the synthetic prompt/video rows have std 0.5: bring the decode
            # tokens' rows to the same scale, otherwise their hidden state is all rounding-sensitive layer output and the bf16
            # cases compare HF-eager vs flash-style rounding rather than the decode logic (the fp32 case pins the logic either way)
            lm.embed_tokens.weight.data.mul_(DECODE_EMBED_SCALE)
        for i in range(decode_steps):
            dec_tokens.append(tok)
            emb = lm.embed_tokens(torch.tensor([[tok]]))
            pid = torch.full((3, 1, 1), T + delta + i, dtype=torch.long)
            o = lm(inputs_embeds=emb.to(dtype), position_ids=pid, past_key_values=cache, use_cache=True,
                   cache_position=torch.tensor([T + i]))
            lg = model.lm_head(o.last_hidden_state[:, -1]).float()[0]
            dec_logits.append(lg.numpy())
            tok = int(torch.argmax(lg))
    for h in hooks:
        h.remove()
    cache_len = np.array([cache.layers[i].keys.shape[2] for i in range(spec.n_layers)], dtype=np.int32)
    meta = dict(name=name, dtype=dtn, frames=frames, grid_h=gh, grid_w=gw, group_size=gs, prefix=prefix, tail=tail,
                top_p=top_p, top_k=top_k, group_tokens=plan.tokens, tail_len=plan.tail_len, embed_seed=4242,
                weight_seed=7, trace=kept_trace)
    if predict_type != "key_norms_small":
        meta["predict_type"] = predict_type
    print(name, plan.tokens, cache_len, float(logits.abs().max()))
    if decode_steps:
        meta["decode_tokens"], meta["rope_delta"], meta["decode_embed_scale"] = dec_tokens, int(delta), DECODE_EMBED_SCALE
        return logits.numpy(), cache_len, meta, np.stack(dec_logits)
    return logits.numpy(), cache_len, meta


def gen_e2e(ref):
    out, meta = {}, []
    for case in E2E_CASES:
        logits, cache_len, m = e2e_case(ref, *case)
        out[f"{case[0]}_logits"], out[f"{case[0]}_cache_len"] = logits, cache_len
        meta.append(m)
    np.savez_compressed(os.path.join(OUT, "gv5_e2e.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "gv5_e2e.json"), "w"), indent=1)


E2E_MODE_CASES = [   # GV5b: the other norm-based predict types end to end (same tiny model / inputs as GV5)
    ("fp32_key_norms", "float32", 8, 4, 6, 4, 5, 7, 0.5, None, "key_norms"),
    ("fp32_vector_norms", "float32", 12, 8, 4, 2, 15, 9, 0.25, None, "vector_norms"),
    ("bf16_vector_norms_small", "bfloat16", 8, 4, 6, 4, 5, 7, 0.5, None, "vector_norms_small"),
    ("bf16_key_norms", "bfloat16", 12, 8, 4, 2, 15, 9, 0.25, None, "key_norms"),
]


def gen_e2e_modes(ref):
    out, meta = {}, []
    for case in E2E_MODE_CASES:
        logits, cache_len, m = e2e_case(ref, *case)
        out[f"{case[0]}_logits"], out[f"{case[0]}_cache_len"] = logits, cache_len
        meta.append(m)
    np.savez_compressed(os.path.join(OUT, "gv5b_e2e_modes.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "gv5b_e2e_modes.json"), "w"), indent=1)


E2E_DECODE_CASES = [E2E_CASES[1], E2E_CASES[4], E2E_CASES[5]]      # fp32_rho05, bf16_rho05, bf16_rho025_g2
DECODE_STEPS = 4
DECODE_EMBED_SCALE = 25.0


def gen_e2e_decode(ref):
    """GV7: the composite reference run continued by greedy decode steps (a10)."""
    out, meta = {}, []
    for case in E2E_DECODE_CASES:
        logits, cache_len, m, dec = e2e_case(ref, *case, decode_steps=DECODE_STEPS)
        out[f"{case[0]}_logits"], out[f"{case[0]}_cache_len"], out[f"{case[0]}_decode_logits"] = logits, cache_len, dec
        meta.append(m)
    np.savez_compressed(os.path.join(OUT, "gv7_e2e_decode.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "gv7_e2e_decode.json"), "w"), indent=1)


def gen_rope_index():
    """GV6: installed transformers' Qwen2-VL get_rope_index (5.15) on grids where it agrees with the
    4.50.0 rule (text after the video resumes at max(position)+1; 5.15 uses max(h,w)//merge, equal
    when t <= max(h,w)//merge)."""
    model, spec, _ = build_hf_tiny(torch.float32)
    vid, vs = model.config.video_token_id, model.config.vision_start_token_id
    rows = []
    for (prefix, t, gh, gw, tail) in [(5, 3, 4, 6, 7), (15, 2, 8, 4, 9), (16, 8, 40, 72, 30), (3, 3, 6, 6, 1), (0, 1, 2, 2, 2)]:
        n_video = t * (gh // 2) * (gw // 2)
        ids = torch.cat([torch.arange(prefix) + 10, torch.full((n_video,), vid), torch.arange(tail) + 10])[None]
        mm = (ids == vid).long() * 2
        pos, delta = model.model.get_rope_index(ids, mm, None, torch.tensor([[t, gh, gw]]), torch.ones_like(ids))
        rows.append(dict(prefix=prefix, t=t, gh=gh, gw=gw, tail=tail, pos_sha=sha(pos[:, 0].numpy().astype(np.int64)),
                         delta=int(delta[0, 0]), last=[int(x) for x in pos[:, 0, -1]]))
        print(rows[-1])
    json.dump(rows, open(os.path.join(OUT, "gv6_rope_index.json"), "w"), indent=1)



# ---------------------------------------------------------------- GV8: full depth at the real 7B dims

DEEP = dict(hidden=3584, n_heads=28, n_kv_heads=4, head_dim=128, intermediate=18944, n_layers=28, vocab=32768)
DEEP_CASE = dict(frames=48, grid_h=16, grid_w=40, group_size=16, prefix=15, tail=30, top_p=0.5, weight_seed=11, embed_seed=12)


def build_hf_text(tspec: dict, seed: int, dtype=torch.bfloat16, attn="sdpa"):
    """Installed transformers' Qwen2-VL text decoder at the given dims with hash-generated weights (O.hashed_text_weights:
    the same bits on any device, so the GPU test regenerates them instead of shipping 15 GB)."""
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextModel
    cfg = Qwen2VLTextConfig(hidden_size=tspec["hidden"], num_attention_heads=tspec["n_heads"], num_key_value_heads=tspec["n_kv_heads"],
                            intermediate_size=tspec["intermediate"], num_hidden_layers=tspec["n_layers"], vocab_size=tspec["vocab"],
                            rms_norm_eps=1e-6, max_position_embeddings=32768, tie_word_embeddings=False,
                            rope_parameters=dict(rope_type="default", mrope_section=[16, 24, 24], rope_theta=1_000_000.0))
    cfg._attn_implementation = attn
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        lm = Qwen2VLTextModel(cfg).eval()
    finally:
        torch.set_default_dtype(prev)
    spec = O.TextSpec(**tspec)
    w = O.hashed_text_weights(spec, seed=seed, dtype=dtype)
    sd = {}
    for k, v in w.items():
        if k == "lm_head.weight":
            continue
        sd[k.replace("q_proj", "self_attn.q_proj").replace("k_proj", "self_attn.k_proj").replace("v_proj", "self_attn.v_proj")
           .replace("o_proj", "self_attn.o_proj")] = v
    missing, unexpected = lm.load_state_dict(sd, strict=False, assign=True)
    assert not unexpected and all("rotary" in m for m in missing), (missing, unexpected)
    return lm, spec, w


def gen_e2e_deep(ref):
    """GV8: 28 layers at the Qwen2-VL-7B dims (d=3584, 28/4 heads, I=18944; vocab cut to 32768), bf16, 3 groups x 1280 tokens
    (+15 prefix on group 0) + 30-token tail, key-norm rho=0.5, through the composite oracle: installed HF decoder (sdpa) + the
    REFERENCE's post_process_kv_cache in a forward hook after every attention, get_top_k_mask_to_predict wrapped to record the mask
    it returns.  Stored: cache lengths, the kept-index list of every (group, layer), the first-token logits."""
    from transformers import DynamicCache
    U, C = ref["utils"], ref["lvu_config"]
    c = DEEP_CASE
    lm, spec, w = build_hf_text(DEEP, c["weight_seed"])
    n_video = (c["frames"] // 2) * (c["grid_h"] // 2) * (c["grid_w"] // 2)
    T = c["prefix"] + n_video + c["tail"]
    plan = O.plan_groups(c["frames"], c["group_size"], c["grid_h"], c["grid_w"], c["prefix"], T)
    pos, delta = O.mrope_positions(c["prefix"], (c["frames"] // 2, c["grid_h"], c["grid_w"]), c["tail"])
    embeds = O.hashed_normal((T, spec.hidden), c["embed_seed"], 0.5)
    cfg = C.LVUConfig(model_name_or_path="x", top_p=c["top_p"])
    layer_cfgs = [C.LVULayerConfig(layer_idx=i, total_layers=spec.n_layers, lvu_config=cfg) for i in range(spec.n_layers)]
    cache = DynamicCache(config=lm.config)
    masks = []
    orig_mask_fn = U.get_top_k_mask_to_predict

    def rec_mask(*a, **kw):
        m = orig_mask_fn(*a, **kw)
        masks.append(torch.nonzero(m[0], as_tuple=True)[0].numpy().astype(np.int16))
        return m
    U.get_top_k_mask_to_predict = rec_mask

    def make_hook(i):
        def hook(mod, args, kwargs, output):
            lay = cache.layers[i]
            with force_stable_argsort():
                res = U.post_process_kv_cache(kwargs["hidden_states"], None, None, None, None, None, (lay.keys, lay.values), layer_cfgs[i])
            lay.keys, lay.values = res[5]
            return output
        return hook
    hooks = [l.self_attn.register_forward_hook(make_hook(i), with_kwargs=True) for i, l in enumerate(lm.layers)]
    segs = plan.tokens + [plan.tail_len]
    post = torch.from_numpy(pos)
    start = 0
    import time
    t0 = time.time()
    with torch.no_grad():
        for gi, n in enumerate(segs):
            if gi == len(segs) - 1:
                cfg.enable = False
            o = lm(inputs_embeds=embeds[None, start:start + n], position_ids=post[:, None, start:start + n], past_key_values=cache,
                   use_cache=True, cache_position=torch.arange(n) + start)
            start += n
            print("segment", gi, n, f"{time.time() - t0:.1f}s", flush=True)
        logits = torch.nn.functional.linear(o.last_hidden_state[0, -1:], w["lm_head.weight"]).float()[0]
    for h in hooks:
        h.remove()
    U.get_top_k_mask_to_predict = orig_mask_fn
    L, G = spec.n_layers, len(plan.tokens)
    assert len(masks) == L * G, len(masks)
    out = {"logits": logits.numpy(), "cache_len": np.array([cache.layers[i].keys.shape[2] for i in range(L)], dtype=np.int32)}
    for gi in range(G):
        for l in range(L):
            out[f"kept_g{gi}_l{l}"] = masks[gi * L + l]
    srt = np.sort(out["logits"])[::-1]
    meta = dict(DEEP_CASE, spec=DEEP, group_tokens=plan.tokens, tail_len=plan.tail_len, rope_delta=int(delta), dtype="bfloat16",
                attn_implementation="sdpa", argmax=int(np.argmax(out["logits"])), top2_margin=float(srt[0] - srt[1]),
                logit_absmax=float(np.abs(out["logits"]).max()))
    np.savez_compressed(os.path.join(OUT, "gv8_deep.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "gv8_deep.json"), "w"), indent=1)
    print(meta)



# ---------------------------------------------------------------- GV9: query-attention-score mode

QUERY_CASES = [  # (Hq, Hkv, n group tokens, m prompt tokens, k)
    (4, 2, 64, 5, 32), (28, 4, 720, 30, 360), (28, 4, 2240, 30, 1120), (8, 1, 960, 24, 480), (12, 2, 333, 1, 100), (28, 4, 5760, 30, 2880),
]


def make_query_case(ci):
    hq, hkv, n, m, k = QUERY_CASES[ci]
    rs = np.random.RandomState(3000 + ci)
    bf = lambda *s, sc=1.0: torch.from_numpy((rs.standard_normal(s) * sc).astype(np.float32)).to(torch.bfloat16)
    return bf(1, hq, n + m, 128, sc=1.3), bf(1, hkv, n + m, 128, sc=1.3), bf(1, hkv, n + m, 128)


def gen_query_scores(ref):
    """GV9: the reference's own LVUCache.update in query-based mode (lvu_cache.py:97-117: prompt K/V stripped, softmax(QK^T) of the
    prompt queries over the group's keys, summed over queries, averaged over heads) on seeded q/k/v, then
    get_top_k_mask_to_predict for both query predict types (utils.py:55-62; argsort forced stable = deployment device)."""
    U, LC = ref["utils"], ref["lvu_cache"]
    out, meta = {}, []
    for ci, (hq, hkv, n, m, k) in enumerate(QUERY_CASES):
        q, kk, vv = make_query_case(ci)
        cache = LC.LVUCache()
        cache.set_prompt_length(m)
        k_all, v_all = cache.update(kk, vv, 0, {"query_states": q})
        assert k_all.shape[2] == n and torch.equal(k_all, kk[:, :, :n])              # prompt K/V never enter the cache
        score = cache.accum_attn_scores[0][-1]
        assert score.shape == (1, n) and score.dtype == torch.bfloat16
        out[f"c{ci}_score_bits"] = O.torch_bf16_to_bits(score[0])
        hid = torch.zeros(1, n, 8)
        for mode in ("query_attention_weights", "query_attention_weights_by_value_norm"):
            with force_stable_argsort():
                mask = U.get_top_k_mask_to_predict(score, k_all, v_all, hid, top_k=k, predict_type=mode)
            idx = torch.nonzero(mask[0], as_tuple=True)[0].numpy().astype(np.int32)
            assert len(idx) == k
            out[f"c{ci}_{mode}"] = idx
        meta.append(dict(case=ci, hq=hq, hkv=hkv, n=n, m=m, k=k, seed=3000 + ci, score_max=float(score.float().max()),
                         distinct_scores=int(len(np.unique(out[f"c{ci}_score_bits"])))))
        print(meta[-1])
    np.savez_compressed(os.path.join(OUT, "gv9_query_scores.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "gv9_query_scores.json"), "w"), indent=1)



# ---------------------------------------------------------------- GV10: vision towers at head_dim 80 from installed transformers

VIT_CASES = {
    # name: (arch, grid_thw, kwargs)
    "qwen2": ("qwen2", (2, 20, 28), dict(depth=3, embed_dim=1280, num_heads=16, mlp_ratio=4.0, out_hidden=256)),
    "qwen25": ("qwen2.5", (2, 20, 28), dict(depth=3, embed_dim=1280, num_heads=16, out_hidden=256, intermediate=3420, window_size=112,
                                           fullatt_blocks=(1,))),
}


def gen_vit_towers():
    """GV10: output of transformers' Qwen2VisionTransformerPretrainedModel / Qwen2_5_VisionTransformerPretrainedModel (fp32 math on
    bf16-rounded, hash-generated weights and pixel rows; 3 blocks at the real width 1280 / 16 heads = head_dim 80, the shape the HIP
    kernels serve; the Qwen2.5 case has ragged border windows and one full-attention block) — the GPU test pins the HIP tower to
    THIS, not to the product's own torch tower."""
    from transformers.models.qwen2_vl.configuration_qwen2_vl import Qwen2VLVisionConfig
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VisionTransformerPretrainedModel
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLVisionConfig
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VisionTransformerPretrainedModel
    out, meta = {}, {}
    for name, (arch, grid, kw) in VIT_CASES.items():
        if arch == "qwen2":
            cfg = Qwen2VLVisionConfig(depth=kw["depth"], embed_dim=kw["embed_dim"], hidden_size=kw["out_hidden"], num_heads=kw["num_heads"],
                                      mlp_ratio=int(kw["mlp_ratio"]), patch_size=14, spatial_merge_size=2, temporal_patch_size=2,
                                      hidden_act="quick_gelu")
            cfg._attn_implementation = "eager"
            m = Qwen2VisionTransformerPretrainedModel(cfg).eval().float()
        else:
            cfg = Qwen2_5_VLVisionConfig(depth=kw["depth"], hidden_size=kw["embed_dim"], intermediate_size=kw["intermediate"], num_heads=kw["num_heads"],
                                         out_hidden_size=kw["out_hidden"], window_size=kw["window_size"], fullatt_block_indexes=list(kw["fullatt_blocks"]),
                                         patch_size=14, spatial_merge_size=2, temporal_patch_size=2, hidden_act="silu")
            cfg._attn_implementation = "eager"
            m = Qwen2_5_VisionTransformerPretrainedModel(cfg).eval().float()
        names_shapes = [(k, list(v.shape)) for k, v in m.state_dict().items()]
        seed = 40 + len(meta)
        sd = O.hashed_state_dict(names_shapes, seed)
        m.load_state_dict({k: v.float() for k, v in sd.items()})
        t, h, w = grid
        pix = O.hashed_normal((t * h * w, 1176), 900 + seed, 1.0)
        with torch.no_grad():
            ref = m(pix.float(), grid_thw=torch.tensor([list(grid)])).pooler_output
        out[f"{name}_out"] = ref.numpy().astype(np.float32)
        meta[name] = dict(arch=arch, grid=list(grid), spec=dict(kw, fullatt_blocks=list(kw.get("fullatt_blocks", ()))), weight_seed=seed,
                          pixel_seed=900 + seed, names_shapes=names_shapes, out_absmax=float(ref.abs().max()))
        print(name, ref.shape, float(ref.abs().max()))
    np.savez_compressed(os.path.join(OUT, "gv10_vit_towers.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "gv10_vit_towers.json"), "w"))



# ---------------------------------------------------------------- GV11: the whole chat_lvu_model — frames -> patches -> ViT -> scatter -> group prefill -> tail -> 4 tokens
# (qwen25_lvu_interleaved.py:733-942 / qwen25_lvu.py:538-761).  Composite oracle: installed transformers' Qwen2-VL / Qwen2.5-VL (vision tower,
# embed_tokens, get_rope_index, decoder, lm_head; seeded tiny checkpoints) driven group by group the way the reference's loop drives it
# (:671-717: ids / positions / pixel rows sliced per group, cache carried), with the REFERENCE's post_process_kv_cache hooked after every
# attention, then the prompt tail without pruning (:737-742) and greedy decode (:744-761).  The checkpoints are NOT stored: the GPU test
# re-creates them from the same seed with the installed transformers and checks `weights_sha256`; frames from `frames_seed`.

from oracle.pipeline_model import (CLIP_MEAN, CLIP_STD, PIPE_CASES, PIPE_DECODE, PIPE_FRAMES, PIPE_GROUP, PIPE_IDS, PIPE_QUESTION, PIPE_SEED,  # noqa: E402
                                   build_hf_pipeline_model, state_sha)


def patch_rows(frames: np.ndarray) -> tuple:
    """uint8 [F, 3, H, W] -> (fp32 rows [t*gh*gw, 3*2*14*14], (t, gh, gw)): CLIP normalisation + the (t, h/2, w/2, 2, 2 | C, T, 14, 14) patch order
    of transformers' Qwen2VLImageProcessor [3P] (what the reference's processor thread produces, qwen25_lvu_interleaved.py:252-271)."""
    F_, C, H, W = frames.shape
    t, gh, gw = F_ // 2, H // 14, W // 14
    x = frames.astype(np.float32) / 255.0
    x = (x - np.array(CLIP_MEAN, np.float32).reshape(1, 3, 1, 1)) / np.array(CLIP_STD, np.float32).reshape(1, 3, 1, 1)
    x = x.reshape(t, 2, C, gh // 2, 2, 14, gw // 2, 2, 14)
    x = x.transpose(0, 3, 6, 4, 7, 2, 1, 5, 8)               # t, hb, wb, hi, wi, C, T, 14, 14
    return np.ascontiguousarray(x.reshape(t * gh * gw, C * 2 * 14 * 14)), (t, gh, gw)


def pipeline_case(ref, family: str, rho: float, dtn: str):
    from transformers import DynamicCache
    from tests.test_processor_seam import installed_qwen2vl_processor
    U, C = ref["utils"], ref["lvu_config"]
    hf, cfg = build_hf_pipeline_model(family)
    wsha = state_sha(hf)
    dtype = getattr(torch, dtn)
    hf = hf.to(dtype)
    fr = PIPE_FRAMES
    frames = np.random.RandomState(fr["seed"]).randint(0, 256, (fr["n"], 3, fr["h"], fr["w"]), dtype=np.uint8)
    rows, grid = patch_rows(frames)
    t, gh, gw = grid
    # the installed image processor agrees with patch_rows on a frame pair made of one image twice (an image IS such a pair to it)
    from PIL import Image
    from transformers import Qwen2VLImageProcessorPil
    ip = Qwen2VLImageProcessorPil()(images=[Image.fromarray(frames[0].transpose(1, 2, 0))], do_resize=False, return_tensors="pt")
    assert tuple(ip["image_grid_thw"][0].tolist()) == (1, gh, gw)
    assert np.allclose(ip["pixel_values"].numpy(), patch_rows(np.stack([frames[0], frames[0]]))[0], atol=1e-5)
    # prompt ids from the INSTALLED Qwen2VLProcessor (chat template + video-pad expansion; qwen25_lvu.py:546-548, 597-604)
    pr = installed_qwen2vl_processor(grid)
    messages = [{"role": "user", "content": [{"type": "video", "video": "v.npy"}, {"type": "text", "text": PIPE_QUESTION}]}]
    text = pr.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
    ids = pr(text=[text], videos=[torch.zeros(fr["n"], 3, 28, 28)], return_tensors="pt")["input_ids"]
    vid = PIPE_IDS["video_token_id"]
    n_video = t * (gh // 2) * (gw // 2)
    assert int((ids == vid).sum()) == n_video
    first = int((ids[0] == vid).nonzero()[0])
    prefix, tail = first, ids.shape[1] - first - n_video
    T = ids.shape[1]
    sample_fps = fr["n"] / (fr["n"] / fr["fps"])
    mm = (ids == vid).long() * 2                                 # transformers 5.x: text 0 / image 1 / video 2
    if family == "qwen2.5-vl":
        extra = dict(second_per_grid_ts=torch.tensor([2.0 / sample_fps]), mm_token_type_ids=mm)
        pos, delta = hf.model.get_rope_index(ids, mm, None, torch.tensor([[t, gh, gw]]), second_per_grid_ts=extra["second_per_grid_ts"],
                                             attention_mask=torch.ones_like(ids))
    else:
        extra = dict(mm_token_type_ids=mm)
        pos, delta = hf.model.get_rope_index(ids, mm, None, torch.tensor([[t, gh, gw]]), torch.ones_like(ids))
    pos = pos[-3:] if pos.shape[0] == 4 else pos                 # (some versions prepend a text-position row)
    lm = hf.model.language_model
    rows_t = torch.from_numpy(rows).to(dtype)
    # group plan (qwen25_lvu.py:623-665): 2 temporal patches per group; group 0 carries the prefix; pixel rows split by frame share
    gs_t = PIPE_GROUP // 2
    per_t_tok, per_t_rows = (gh // 2) * (gw // 2), gh * gw
    groups, tok0, row0 = [], 0, 0
    for g0 in range(0, t, gs_t):
        tt = min(gs_t, t - g0)
        n = tt * per_t_tok + (prefix if g0 == 0 else 0)
        groups.append((tok0, n, row0, tt))
        tok0 += n; row0 += tt * per_t_rows
    lcfg = C.LVUConfig(model_name_or_path="x", top_p=rho if rho < 1.0 else None, top_k_predict_type="key_norms_small")
    layer_cfgs = [C.LVULayerConfig(layer_idx=i, total_layers=len(lm.layers), lvu_config=lcfg) for i in range(len(lm.layers))]
    cache = DynamicCache(config=hf.config.get_text_config())
    trace = []

    def make_hook(i):
        def hook(mod, args, kwargs, output):
            lay = cache.layers[i]
            before = lay.keys.shape[2]
            with force_stable_argsort():
                res = U.post_process_kv_cache(kwargs["hidden_states"], None, None, None, None, None, (lay.keys, lay.values), layer_cfgs[i])
            lay.keys, lay.values = res[5]
            trace.append((i, before, lay.keys.shape[2]))
            return output
        return hook
    hooks = [l.self_attn.register_forward_hook(make_hook(i), with_kwargs=True) for i, l in enumerate(lm.layers)]
    vit = hf.model.visual
    with torch.no_grad():
        for (s0, n, r0, tt) in groups:
            gi = ids[:, s0:s0 + n]
            emb = lm.embed_tokens(gi)
            v = vit(rows_t[r0:r0 + tt * per_t_rows], grid_thw=torch.tensor([[tt, gh, gw]]))
            v = v.pooler_output if hasattr(v, "pooler_output") else v
            v = v[0] if isinstance(v, (tuple, list)) else v
            emb = emb.masked_scatter((gi == vid)[..., None].expand_as(emb), v.to(emb.dtype))           # qwen25_lvu.py [3P] forward: masked_scatter of the video embeds
            o = lm(inputs_embeds=emb, position_ids=pos[:, :, s0:s0 + n], past_key_values=cache, use_cache=True, cache_position=torch.arange(n) + s0)
        lcfg.enable = False                                          # qwen25_lvu.py:737-738: the prompt tail is not pruned
        s0 = T - tail
        o = lm(inputs_embeds=lm.embed_tokens(ids[:, s0:]), position_ids=pos[:, :, s0:], past_key_values=cache, use_cache=True,
               cache_position=torch.arange(tail) + s0)
        cache_len = [int(cache.layers[i].keys.shape[2]) for i in range(len(lm.layers))]
        logits = hf.lm_head(o.last_hidden_state[:, -1]).float()[0]
        step_logits, toks = [logits.numpy()], []
        tok = int(torch.argmax(logits))
        for i in range(PIPE_DECODE - 1):
            toks.append(tok)
            pid = torch.full((3, 1, 1), T + int(delta[0, 0]) + i, dtype=torch.long)
            o = lm(inputs_embeds=lm.embed_tokens(torch.tensor([[tok]])), position_ids=pid, past_key_values=cache, use_cache=True,
                   cache_position=torch.tensor([T + i]))
            lg = hf.lm_head(o.last_hidden_state[:, -1]).float()[0]
            step_logits.append(lg.numpy())
            tok = int(torch.argmax(lg))
        toks.append(tok)
    for h in hooks:
        h.remove()
    whole = None
    if rho >= 1.0 and dtn == "float32":
        # pins the composite (group slicing, positions, scatter, cache carry) to ONE forward of the installed model over the whole prompt
        with torch.no_grad():
            out = hf(input_ids=ids, pixel_values_videos=rows_t, video_grid_thw=torch.tensor([[t, gh, gw]]), attention_mask=torch.ones_like(ids), **extra)
        whole = float((out.logits[0, -1].float() - logits).abs().max())
        assert whole < 2e-4, whole
    meta = dict(family=family, rho=rho, dtype=dtn, weights_sha256=wsha, prefix_ids=ids[0, :prefix].tolist(), tail_ids=ids[0, T - tail:].tolist(),
                n_video=n_video, grid=[t, gh, gw], group_tokens=[g[1] for g in groups], tail_len=tail, rope_delta=int(delta[0, 0]),
                cache_len=cache_len, tokens=toks, trace=trace, composite_vs_whole_forward_max_abs=whole,
                margins=[float(np.sort(l)[-1] - np.sort(l)[-2]) for l in step_logits], max_abs_logit=float(np.abs(step_logits[0]).max()))
    return meta, np.stack(step_logits)


def gen_e2e_pipeline(ref):
    """GV11.  fp32 run = the reference numbers; bf16 run of the same composite = how far bf16 arithmetic alone moves them (the GPU test's
    tolerance is stated against that distance)."""
    out, metas = {}, []
    fr = PIPE_FRAMES
    frames = np.random.RandomState(fr["seed"]).randint(0, 256, (fr["n"], 3, fr["h"], fr["w"]), dtype=np.uint8)
    for family, rho in PIPE_CASES:
        name = f"{family}_rho{rho}"
        m32, l32 = pipeline_case(ref, family, rho, "float32")
        m16, l16 = pipeline_case(ref, family, rho, "bfloat16")
        m32["bf16_oracle"] = dict(tokens=m16["tokens"], cache_len=m16["cache_len"], first_logits_max_abs_diff=float(np.abs(l16[0] - l32[0]).max()),
                                  first_logits_cosine=float(np.dot(l16[0], l32[0]) / (np.linalg.norm(l16[0]) * np.linalg.norm(l32[0]))))
        m32["name"] = name
        out[f"{name}_logits"] = l32.astype(np.float32)
        out[f"{name}_logits_bf16_oracle"] = l16.astype(np.float32)
        metas.append(m32)
        print(name, "tokens", m32["tokens"], "bf16 oracle tokens", m16["tokens"], "cache", m32["cache_len"], "margins", [round(x, 3) for x in m32["margins"]],
              "bf16-vs-fp32 first logits", round(m32["bf16_oracle"]["first_logits_max_abs_diff"], 4), "whole-forward", m32["composite_vs_whole_forward_max_abs"])
    rec = dict(frames=dict(fr, sha256=sha(frames)), question=PIPE_QUESTION, video_group_size=PIPE_GROUP, num_frames=fr["n"], decode_steps=PIPE_DECODE,
               head_scale=16.0, init_seed=PIPE_SEED, cases=metas)
    np.savez_compressed(os.path.join(OUT, "gv11_e2e_pipeline.npz"), **out)
    json.dump(rec, open(os.path.join(OUT, "gv11_e2e_pipeline.json"), "w"), indent=1)


# ---------------------------------------------------------------- GV4: frame count + frame size from the video entry
NFRAMES_ELES = [{}, {"fps": 1}, {"fps": 2}, {"fps": 0.5}, {"fps": 4.0}, {"fps": 2, "min_frames": 16}, {"fps": 2, "max_frames": 128},
                {"fps": 2, "min_frames": 7, "max_frames": 65}, {"fps": 1, "max_frames": 768}, {"nframes": 64}, {"nframes": 7}, {"nframes": 5},
                {"nframes": 7200}, {"nframes": 3}, {"nframes": 1}, {"nframes": 128}, {"fps": 2, "nframes": 64}]
NFRAMES_VIDEOS = [(4, 2.0), (37, 24.0), (64, 2.0), (257, 29.97), (1800, 30.0), (7200, 2.0), (28800, 8.0), (28416, 24.0), (86313, 23.976),
                  (216000, 60.0), (2, 30.0), (1, 25.0)]
SIZE_ELES = [{}, {"max_pixels": 10 ** 7}, {"max_pixels": 392 * 560}, {"max_pixels": 200 * 28 * 28}, {"max_pixels": 360 * 420},
             {"min_pixels": 256 * 28 * 28}, {"min_pixels": 16 * 28 * 28}, {"min_pixels": 256 * 28 * 28, "max_pixels": 300 * 28 * 28},
             {"min_pixels": 900 * 28 * 28}, {"total_pixels": 7200 * 392 * 560 // 2}, {"total_pixels": 7200 * 392 * 560 // 2 + 7200 * 14 * 28},
             {"total_pixels": 128000 * 28 * 28 * 0.9}, {"total_pixels": 1000 * 28 * 28}, {"total_pixels": 10 ** 9, "max_pixels": 400 * 28 * 28},
             {"resized_height": 280, "resized_width": 420}, {"resized_height": 300, "resized_width": 500, "max_pixels": 10 ** 7},
             {"resized_height": 280}]
SIZE_VIDEOS = [(nf, h, w) for nf in (2, 16, 64, 256, 512, 768, 7200, 14400)
               for (h, w) in ((1080, 1920), (720, 1280), (392, 560), (480, 640), (2160, 3840), (360, 640), (240, 320), (1920, 1080), (300, 2000))]
# (sizes that smart_resize would shrink below ONE 28-pixel patch row are left out: transformers' smart_resize floors them to 28 where
#  qwen-vl-utils 0.0.10 returns 0 — a frame the reference cannot process either)


def extract_video_planning():
    """AST-extracts, from the reference's plugin sources (build container only; nothing is copied into the repo), what decides how many
    frames are sampled and at what size: the module-level `smart_nframes` (qwen25_lvu.py:402-442) and the budget / resize statements of
    `fetch_video` (qwen25_lvu.py:350-372; qwen25_lvu_interleaved.py:416-436) — and compiles them against the [3P] names they use:
    qwen-vl-utils' published constants and one-line helpers, and the installed transformers' `smart_resize` (the same arithmetic as
    qwen-vl-utils 0.0.10 for every non-degenerate size) under qwen-vl-utils' default pixel limits.  Returns callables + a warning log."""
    import ast
    import math
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import smart_resize as hf_smart_resize

    warned = []

    class Log:
        def warning(self, msg):
            warned.append(msg)

        info = debug = warning

    def smart_resize(height, width, factor=28, min_pixels=4 * 28 * 28, max_pixels=16384 * 28 * 28):     # qwen-vl-utils defaults MIN/MAX_PIXELS
        return hf_smart_resize(height, width, factor, min_pixels, max_pixels)

    env = {"IMAGE_FACTOR": 28, "FRAME_FACTOR": 2, "FPS": 2.0, "FPS_MIN_FRAMES": 4, "VIDEO_MIN_PIXELS": 128 * 28 * 28,
           "VIDEO_MAX_PIXELS": 768 * 28 * 28, "VIDEO_TOTAL_PIXELS": 24576 * 28 * 28, "logger": Log(), "smart_resize": smart_resize,
           "round_by_factor": lambda n, f: round(n / f) * f, "ceil_by_factor": lambda n, f: math.ceil(n / f) * f,
           "floor_by_factor": lambda n, f: math.floor(n / f) * f}
    out = {}
    dumps = []
    for fname in ("qwen25_lvu.py", "qwen25_lvu_interleaved.py"):
        tree = ast.parse(open(os.path.join(REF, "lvu", "models", fname)).read())
        g = dict(env)
        for node in tree.body:                                      # the reference's own override: FPS_MAX_FRAMES = 100_000 (qwen25_lvu.py:27)
            if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", None) == "FPS_MAX_FRAMES":
                exec(compile(ast.Module([node], []), fname, "exec"), g)
        assert g["FPS_MAX_FRAMES"] == 100_000                       # both plugins carry the override (interleaved:32)
        fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "smart_nframes")
        dumps.append(ast.dump(fn))
        exec(compile(ast.Module([fn], []), fname, "exec"), g)
        fv = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "fetch_video")
        branch = fv.body[0]                                         # `if isinstance(ele["video"], str):`
        assert isinstance(branch, ast.If)
        names = {"total_pixels", "min_pixels", "max_pixels", "max_pixels_supposed"}
        keep = []
        for st in branch.body:
            src = ast.unparse(st)
            if isinstance(st, ast.Assign) and all(isinstance(t, ast.Name) and t.id in names for t in st.targets):
                keep.append(st)
            elif isinstance(st, ast.If) and ("max_pixels_supposed" in src.split(":")[0] or "resized_height" in src.split(":")[0]):
                keep.append(st)
        assert len(keep) == 7, [ast.unparse(k)[:50] for k in keep]   # 5 assignments + the warning + the resize decision
        ret = ast.parse("return (resized_height, resized_width, max_pixels)").body[0]
        fdef = ast.parse("def frame_size(ele, nframes, height, width, image_factor=28, video_reader_backend='decord'): pass").body[0]
        fdef.body = keep + [ret]
        mod = ast.fix_missing_locations(ast.Module([fdef], []))
        exec(compile(mod, fname + ":fetch_video", "exec"), g)
        out[fname] = (g["smart_nframes"], g["frame_size"], g["FPS_MAX_FRAMES"])
    assert dumps[0] == dumps[1], "the two plugins' smart_nframes differ"
    return out, warned


def gen_video_plan():
    """GV4: (video entry, total_frames, video_fps) -> nframes and (video entry, nframes, source size) -> max_pixels, (H, W), computed by the
    reference's own statements (extract_video_planning)."""
    fns, warned = extract_video_planning()
    rec = {"note": "outputs of the reference's smart_nframes and of fetch_video's budget/resize statements, AST-extracted from "
                   "lvu/models/qwen25_lvu.py (FPS_MAX_FRAMES = 100_000, :27) and checked equal on qwen25_lvu_interleaved.py's twins; [3P] names: "
                   "qwen-vl-utils 0.0.10 constants, transformers' smart_resize", "nframes": [], "frame_size": []}
    sn, fs, _ = fns["qwen25_lvu.py"]
    sn_i, fs_i, cap_i = fns["qwen25_lvu_interleaved.py"]
    for total, vfps in NFRAMES_VIDEOS:
        for ele in NFRAMES_ELES:
            def run(f):
                try:
                    return {"nframes": f(dict(ele), total, vfps)}
                except (ValueError, AssertionError) as e:
                    return {"raises": type(e).__name__, "message": str(e)}
            r = run(sn)
            assert run(sn_i) == r
            rec["nframes"].append({"ele": ele, "total_frames": total, "video_fps": vfps, **r})
    for nf, h, w in SIZE_VIDEOS:
        for ele in SIZE_ELES:
            del warned[:]
            rh, rw, mp = fs(dict(ele), nf, h, w)
            w1 = len(warned) > 0
            assert fs_i(dict(ele), nf, h, w) == (rh, rw, mp)
            rec["frame_size"].append({"ele": ele, "nframes": nf, "height": h, "width": w, "resized": [rh, rw], "max_pixels": mp, "warned": w1})
    # the two divergences the round-3 review measured
    assert fs({"max_pixels": 10 ** 7}, 64, 1080, 1920)[:2] == (560, 1008) and fs({"max_pixels": 392 * 560}, 7200, 392, 560)[:2] == (252, 364)
    with open(os.path.join(OUT, "gv4_video_plan.json"), "w") as f:
        json.dump(rec, f, indent=0)
    print("gv4_video_plan.json:", len(rec["nframes"]), "frame-count cases,", len(rec["frame_size"]), "frame-size cases")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    which = sys.argv[1:] or ["select", "modes", "effk", "compact", "e2e", "e2e_modes", "e2e_decode", "rope", "query", "vit", "video_plan", "pipeline"]
    if "select" in which: gen_select(ref)
    if "modes" in which: gen_select_modes(ref)
    if "effk" in which: gen_effective_k(ref)
    if "compact" in which: gen_compaction(ref)
    if "e2e" in which: gen_e2e(ref)
    if "e2e_modes" in which: gen_e2e_modes(ref)
    if "e2e_decode" in which: gen_e2e_decode(ref)
    if "rope" in which: gen_rope_index()
    if "query" in which: gen_query_scores(ref)
    if "vit" in which: gen_vit_towers()
    if "video_plan" in which: gen_video_plan()
    if "pipeline" in which: gen_e2e_pipeline(ref)
    if "deep" in which: gen_e2e_deep(ref)       # ~20 min of CPU and 40 GB of RAM: not in the default list
