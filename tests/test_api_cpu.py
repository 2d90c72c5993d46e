"""CPU tests of the drop-in API surface (lvu.LVU / LVUConfig / plugin registry), the video->token pipeline with the
oracle-backed ops double, and the tensor-parallel path over gloo (world_size 2)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import qp_oracle as O
from tests.oracle_ops import OracleOps
from tests.test_engine_host import make_case


def test_lvu_surface_and_registry():
    import lvu
    from quickvideo_amd.models import lvu_chat_model_map, lvu_init_model_map, lvu_run_model_map
    assert set(lvu_init_model_map) == set(lvu_run_model_map) >= {"qwen2vl_mi355x", "qwen2vl_mi355x_sequential"}
    assert "qwen2vl_mi355x" in lvu_chat_model_map
    from quickvideo_amd.lvu import load_native_model
    m = load_native_model("synthetic:tiny", device="cpu")
    with pytest.raises(ValueError, match="not supported"):
        lvu.LVU(lvu.LVUConfig("synthetic:tiny", model_type="no_such_plugin"), model=m)       # lvu.py:33-34
    with pytest.raises(ValueError):
        load_native_model("Qwen/NoSuchModel", device="cpu")
    # the reference's own registry keys (lvu/models/*.py file stems) run unchanged: overlapped <-> interleaved, the other two sequential
    from quickvideo_amd.models import REFERENCE_ALIASES
    assert set(REFERENCE_ALIASES) == {"qwen25_lvu_interleaved", "qwen25_lvu", "qwen25_vl"}
    for alias, target in REFERENCE_ALIASES.items():
        assert lvu_run_model_map[alias] is lvu_run_model_map[target] and lvu_init_model_map[alias] is lvu_init_model_map[target]
        assert lvu_chat_model_map[alias] is lvu_chat_model_map[target]
    ref_style = lvu.LVU(lvu.LVUConfig("synthetic:tiny", model_type="qwen25_lvu_interleaved", top_p=0.5, video_group_size=4, num_frames=8), model=m)
    assert ref_style.run_model_func.__func__ is lvu_run_model_map["qwen2vl_mi355x"]
    obj = lvu.LVU(lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=4, num_frames=8), model=m)
    assert callable(obj.generate) and callable(obj.chat) and obj.model is m


@pytest.mark.parametrize("model_type", ["qwen2vl_mi355x", "qwen2vl_mi355x_sequential"])
def test_generate_end_to_end_cpu(model_type, capsys):
    import lvu
    from quickvideo_amd.lvu import load_native_model
    m = load_native_model("synthetic:tiny", device="cpu")
    cfg = lvu.LVUConfig("synthetic:tiny", model_type=model_type, top_p=0.5, video_group_size=4, num_frames=8)
    obj = lvu.LVU(cfg, model=m)
    obj._ops = OracleOps()
    video = "synthetic://?frames=40&h=56&w=84&seed=3"
    out = obj.generate("What happens in the video?", video, max_new_tokens=3)
    assert isinstance(out, list) and len(out) == 1 and out[0].count("<tok_") == 3
    t = obj._pipeline.last_timings
    assert t.groups == 2 and t.tokens > 0 and t.ttft > 0
    printed = capsys.readouterr().out
    assert "total time spent on prefill was" in printed and "e2e" in printed
    # chat() with the message run_lvu_model builds (the video entry carries `nframes`, qwen25_lvu.py:504-536) gives the same answer
    msgs = [{"role": "user", "content": [{"type": "video", "video": video, "nframes": 8}, {"type": "text", "text": "What happens in the video?"}]}]
    assert obj.chat(msgs, max_new_tokens=3) == out
    # ... and chat() reads ONLY the video entry, like the reference's chat_lvu_model: a bare entry is sampled at qwen-vl-utils' default
    # 2 fps (40 frames at 2 fps -> all 40), whatever LVUConfig.num_frames says; per-entry fps / max_frames are honoured
    from quickvideo_amd.frames import open_video
    bare = [{"role": "user", "content": [{"type": "video", "video": video}, {"type": "text", "text": "q"}]}]
    assert obj._pipeline.plan(open_video(video), bare)["nframes"] == 40
    e2 = [{"role": "user", "content": [{"type": "video", "video": video, "fps": 1.0, "max_frames": 12}, {"type": "text", "text": "q"}]}]
    assert obj._pipeline.plan(open_video(video), e2)["nframes"] == 12
    e3 = [{"role": "user", "content": [{"type": "video", "video": video, "nframes": 8, "resized_height": 60, "resized_width": 110}, {"type": "text", "text": "q"}]}]
    P3 = obj._pipeline.plan(open_video(video), e3)
    assert (P3["nframes"], P3["H"], P3["W"]) == (8, 56, 112)
    # the sequential and overlapped plugins agree
    other = "qwen2vl_mi355x_sequential" if model_type == "qwen2vl_mi355x" else "qwen2vl_mi355x"
    obj2 = lvu.LVU(lvu.LVUConfig("synthetic:tiny", model_type=other, top_p=0.5, video_group_size=4, num_frames=8), model=m)
    obj2._ops = OracleOps()
    assert obj2.generate("What happens in the video?", video, max_new_tokens=3) == out


@pytest.mark.parametrize("predict_type", ["key_norms_small", "query_attention_weights"])
def test_pipeline_matches_oracle_first_token(predict_type):
    """Pipeline (frames -> patchify -> ViT -> scatter -> group prefill) vs the oracle's group_prefill fed with the same
    ViT features: same first token and logits (same math through the ops double).  Also in the query-based mode, where the
    pipeline appends the prompt rows (and the next tail_len positions) to every group (qwen25_lvu.py:684-689)."""
    import lvu
    from quickvideo_amd.frames import open_video
    from quickvideo_amd.lvu import load_native_model
    from quickvideo_amd.pipeline import PrefillPipeline
    from quickvideo_amd.processor import SyntheticProcessor
    from quickvideo_amd.vit import VisionTower, patchify_frames
    m = load_native_model("synthetic:tiny", device="cpu", seed=3)
    cfg = lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=4, num_frames=8, top_k_predict_type=predict_type)
    pipe = PrefillPipeline(m, cfg, SyntheticProcessor(m.spec), ops=OracleOps())
    video = "synthetic://?frames=16&h=56&w=84&seed=5&pattern=gradient"
    ids = pipe.generate("Describe the scene", video, max_new_tokens=1)
    r = open_video(video)
    P = pipe.plan(r, "Describe the scene")
    r.height, r.width = P["H"], P["W"]; r.frame_iter = P["nframes"]; r.process(P["idx"])
    frames = next(r)
    rows, grid = patchify_frames(frames, m.vision.spec, torch.bfloat16)
    feats = VisionTower(m.vision).forward(rows, grid)
    emb = torch.cat([m.text.embed[torch.tensor(P["prompt"].prefix_ids)], feats, m.text.embed[torch.tensor(P["prompt"].tail_ids)]], 0)
    w = {"embed_tokens.weight": m.text.embed, "norm.weight": m.text.norm, "lm_head.weight": m.text.lm_head}
    s = m.spec
    for l, lw in enumerate(m.text.layers):
        p = f"layers.{l}."
        qd, kd = s.q_dim, s.kv_dim
        w[p + "input_layernorm.weight"], w[p + "post_attention_layernorm.weight"] = lw.ln1, lw.ln2
        w[p + "q_proj.weight"], w[p + "k_proj.weight"], w[p + "v_proj.weight"] = lw.w_qkv[:qd], lw.w_qkv[qd:qd + kd], lw.w_qkv[qd + kd:]
        w[p + "q_proj.bias"], w[p + "k_proj.bias"], w[p + "v_proj.bias"] = lw.b_qkv[:qd], lw.b_qkv[qd:qd + kd], lw.b_qkv[qd + kd:]
        w[p + "o_proj.weight"], w[p + "mlp.gate_proj.weight"], w[p + "mlp.up_proj.weight"] = lw.w_o, lw.w_gate_up[:s.intermediate], lw.w_gate_up[s.intermediate:]
        w[p + "mlp.down_proj.weight"] = lw.w_down
    so = O.TextSpec(hidden=s.hidden, n_heads=s.n_heads, n_kv_heads=s.n_kv_heads, head_dim=s.head_dim, intermediate=s.intermediate,
                    n_layers=s.n_layers, vocab=s.vocab)
    ref = O.group_prefill(w, so, emb, P["pos"], P["plan"].tokens, O.PruneCfg(top_p=0.5, top_k_predict_type=predict_type))
    assert int(torch.argmax(ref["logits"])) == ids[0]
    assert m.engine.arena.len == ref["cache_len"]


def test_local_attention_off_engine():
    """adaptive_local_attention=False: groups do not see earlier groups (qwen25_lvu.py:700-714); the tail sees everything."""
    from quickvideo_amd import planner
    from quickvideo_amd.engine import QuickPrefillEngine
    from quickvideo_amd.lvu_config import LVUConfig
    from quickvideo_amd.spec import TINY
    from quickvideo_amd.weights import DecoderWeights
    spec_o, w, plan, pos, delta, embeds = make_case(8, 4, 6, 4, 5, 7)
    cfg = LVUConfig("x", top_p=0.5, video_group_size=4, adaptive_local_attention=False)
    eng = QuickPrefillEngine(DecoderWeights.from_named(TINY, w, "cpu"), cfg, capacity=64, max_group_tokens=32, device="cpu", ops=OracleOps())
    post = torch.from_numpy(pos)
    eng.prefill_group(embeds[:17], post[:, :17]); eng.prefill_group(embeds[17:29], post[:, 17:29])
    k_after = [eng.arena.k(l)[:, :eng.arena.len[l]].clone() for l in range(3)]
    # group 1 prefetched alone into an empty engine must produce the same K rows for layer 0..2
    eng2 = QuickPrefillEngine(DecoderWeights.from_named(TINY, w, "cpu"), cfg, capacity=64, max_group_tokens=32, device="cpu", ops=OracleOps())
    eng2.prefill_group(embeds[17:29], post[:, 17:29])
    for l in range(3):
        assert torch.equal(k_after[l][:, -eng2.arena.len[l]:], eng2.arena.k(l)[:, :eng2.arena.len[l]])


# ---------------------------------------------------------------- tensor parallel over gloo (world_size 2)
def _tp_worker(rank, world, port, ret, hq=4, hkv=2):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from quickvideo_amd.engine import QuickPrefillEngine
    from quickvideo_amd.lvu_config import LVUConfig
    from quickvideo_amd.spec import TextSpec
    from quickvideo_amd.weights import DecoderWeights
    so = O.TextSpec(hidden=512, n_heads=hq, n_kv_heads=hkv, head_dim=128, intermediate=512, n_layers=2, vocab=128)
    spec = TextSpec(hidden=512, n_heads=hq, n_kv_heads=hkv, head_dim=128, intermediate=512, n_layers=2, vocab=128)
    w = {k: v.to(torch.bfloat16) for k, v in O.synthetic_text_weights(so, seed=5, norm_jitter=0.1).items()}
    rs = np.random.RandomState(9)
    T, groups = 60, [24, 24]
    embeds = torch.from_numpy(rs.standard_normal((T, 512)).astype(np.float32) * 0.5).to(torch.bfloat16)
    pos = np.tile(np.arange(T, dtype=np.int64), (3, 1))
    cfg = LVUConfig("x", top_p=0.5, video_group_size=4)
    dw = DecoderWeights.from_named(spec, w, "cpu", tp_rank=rank, tp_size=world)
    eng = QuickPrefillEngine(dw, cfg, capacity=T + 4, max_group_tokens=24, device="cpu", ops=OracleOps(), tp_group=dist.group.WORLD)
    eng.kept_trace = []
    post = torch.from_numpy(pos)
    st = 0
    for n in groups:
        eng.prefill_group(embeds[st:st + n], post[:, st:st + n]); st += n
    logits = eng.prefill_tail(embeds[st:], post[:, st:])
    kept = [None if k is None else k.numpy().copy() for _, k in eng.kept_trace]
    if rank == 0:
        ref = O.group_prefill(w, so, embeds, pos, groups, O.PruneCfg(top_p=0.5))
        ret["ref_logits"], ret["ref_len"] = ref["logits"].numpy(), ref["cache_len"]
        ret["ref_kept"] = [k for g in ref["kept"] for k in g]
    ret[f"logits{rank}"], ret[f"kept{rank}"], ret[f"len{rank}"] = logits.numpy(), kept, list(eng.arena.len)
    ret[f"heads{rank}"] = (eng.hq, eng.hkv, eng.li)
    dist.destroy_process_group()


def _tp_chunk_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from quickvideo_amd.engine import QuickPrefillEngine
    from quickvideo_amd.lvu_config import LVUConfig
    from quickvideo_amd.spec import TextSpec
    from quickvideo_amd.weights import DecoderWeights
    hq, hkv = 4, 2
    so = O.TextSpec(hidden=256, n_heads=hq, n_kv_heads=hkv, head_dim=128, intermediate=256, n_layers=2, vocab=64)
    spec = TextSpec(hidden=256, n_heads=hq, n_kv_heads=hkv, head_dim=128, intermediate=256, n_layers=2, vocab=64)
    w = {k: v.to(torch.bfloat16) for k, v in O.synthetic_text_weights(so, seed=6, norm_jitter=0.1).items()}
    rs = np.random.RandomState(10)
    n, T = 1100, 1108                                     # one group of 1100 rows: 2 (4) row blocks of the projections, ragged last block
    embeds = torch.from_numpy(rs.standard_normal((T, 256)).astype(np.float32) * 0.5).to(torch.bfloat16)
    post = torch.from_numpy(np.tile(np.arange(T, dtype=np.int64), (3, 1)))
    cfg = LVUConfig("x", top_p=0.5, video_group_size=4)
    dw = DecoderWeights.from_named(spec, w, "cpu", tp_rank=rank, tp_size=world)
    for chunks in (1, 2, 4):
        os.environ["QP_TP_CHUNKS"] = str(chunks)
        eng = QuickPrefillEngine(dw, cfg, capacity=T + 4, max_group_tokens=n, device="cpu", ops=OracleOps(), tp_group=dist.group.WORLD)
        assert eng.tp_chunks == chunks
        eng.kept_trace = []
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda t, *a, **kw: (calls.append((tuple(t.shape), kw.get("async_op", False))), real(t, *a, **kw))[1]
        try:
            eng.prefill_group(embeds[:n], post[:, :n])
            logits = eng.prefill_tail(embeds[n:], post[:, n:])
        finally:
            dist.all_reduce = real
        ret[f"c{chunks}_{rank}"] = (logits.numpy().copy(), [k.numpy().copy() for _, k in eng.kept_trace if k is not None], calls)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_tensor_parallel_allreduce_row_blocks_equal_the_unsplit_form(world):
    """VERDICT r3 #5a: o_proj / down_proj + all-reduce in row blocks, block i's all-reduce asynchronous beside block i+1's GEMM
    (QP_TP_CHUNKS=2, the default; 4) vs the unsplit form (QP_TP_CHUNKS=1).  Two ranks: BIT-identical logits and kept lists (a + b is
    commutative; rows are independent in the GEMM).  Four ranks: the ring adds the four partials of an element in an order that depends
    on the element's place in the message, so the first layer's kept list must agree exactly and the rest (second layer's kept list,
    logits) to bf16 noise — the same holds between any two message sizes of the unsplit form."""
    ret = mp.Manager().dict()
    mp.spawn(_tp_chunk_worker, args=(world, 31500 + os.getpid() % 2000 + world, ret), nprocs=world, join=True)
    for r in range(world):
        l1, k1, c1 = ret[f"c1_{r}"]
        assert all(not a for _, a in c1) and sum(1 for s_, _ in c1 if s_ == (1100, 256)) == 4           # 2 layers x (o, down), whole, blocking
        for chunks in (2, 4):
            l, k, c = ret[f"c{chunks}_{r}"]
            big = [(s_, a) for s_, a in c if s_[0] > 8]
            assert len(big) == 4 * chunks and all(a for _, a in big) and sum(s_[0] for s_, _ in big) == 4 * 1100   # asynchronous row blocks
            assert len(k) == len(k1) and np.array_equal(k[0], k1[0])                      # layer 0 sees identical inputs on every form
            if world == 2:
                assert all(np.array_equal(a, b) for a, b in zip(k, k1))
                assert np.array_equal(l, l1), float(np.max(np.abs(l - l1)))
            else:                                                                        # layer 1 sits behind a re-ordered bf16 sum
                assert all(len(set(a.tolist()) & set(b.tolist())) / len(a) >= 0.97 for a, b in zip(k, k1))
                assert np.max(np.abs(l - l1)) <= 4e-2
        assert np.array_equal(ret[f"c2_{r}"][0], ret["c2_0"][0])                                          # ranks agree with each other


def test_tensor_parallel_gloo_world2():
    port = 29500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_tp_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["heads0"] == ret["heads1"] == (2, 1, 256)                      # 4 q heads / 2 kv heads / I=512 split in two
    assert ret["len0"] == ret["len1"] == ret["ref_len"]
    # every rank derives the IDENTICAL kept-index lists (all-gathered per-head partials, fixed head order)
    for a, b in zip(ret["kept0"], ret["kept1"]):
        assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))
    # ... and they match the single-device oracle up to near-ties from the bf16 all-reduce rounding
    tot = same = 0
    for a, r in zip(ret["kept0"], ret["ref_kept"]):
        if r is not None:
            tot += len(r); same += len(set(a.tolist()) & set(r.tolist()))
    assert same / tot >= 0.9
    assert np.array_equal(ret["logits0"], ret["logits1"])
    ref = ret["ref_logits"]
    assert np.max(np.abs(ret["logits0"] - ref)) <= 4e-2


def test_tensor_parallel_gloo_world4_replicated_kv_heads():
    """tp > n_kv_heads: every kv head replicated on 2 ranks, its 3 q heads dealt 2 + 1(+1 zero pad head) — the layout
    Qwen2-VL-7B (28 q / 4 kv heads) needs at TP=8."""
    from quickvideo_amd.weights import tp_head_partition
    assert [tp_head_partition(28, 4, r, 8)[0] for r in (0, 1, 7)] == [[0, 1, 2, 3], [4, 5, 6, -1], [25, 26, 27, -1]]
    assert tp_head_partition(64, 8, 3, 8) == (list(range(24, 32)), 3, 1)
    port = 31500 + os.getpid() % 2000
    ret = mp.Manager().dict()
    mp.spawn(_tp_worker, args=(4, port, ret, 6, 2), nprocs=4, join=True)
    assert ret["heads0"] == ret["heads3"] == (2, 1, 128)
    assert ret["len0"] == ret["len1"] == ret["len2"] == ret["len3"] == ret["ref_len"]
    for r in (1, 2, 3):
        for a, b in zip(ret["kept0"], ret[f"kept{r}"]):
            assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))
        assert np.array_equal(ret["logits0"], ret[f"logits{r}"])
    assert np.max(np.abs(ret["logits0"] - ret["ref_logits"])) <= 4e-2


def test_tensor_parallel_gloo_world8_one_kv_head_per_rank():
    """The Qwen2-VL-72B layout of BASELINE.json configs[4] at tiny width: 8 kv heads over 8 ranks (one each, 2 q heads per rank),
    MLP columns / 8; every rank must derive the identical kept-index lists from the all-gathered per-head key sums (fixed head
    order) and the identical logits; cache lengths and logits vs the single-process oracle."""
    port = 33500 + os.getpid() % 2000
    ret = mp.Manager().dict()
    mp.spawn(_tp_worker, args=(8, port, ret, 16, 8), nprocs=8, join=True)
    assert all(ret[f"heads{r}"] == (2, 1, 64) for r in range(8))
    assert all(ret[f"len{r}"] == ret["ref_len"] for r in range(8))
    for r in range(1, 8):
        for a, b in zip(ret["kept0"], ret[f"kept{r}"]):
            assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))
        assert np.array_equal(ret["logits0"], ret[f"logits{r}"])
    assert np.max(np.abs(ret["logits0"] - ret["ref_logits"])) <= 4e-2


def test_load_hf_checkpoint_directory(tmp_path):
    """Local HF-layout checkpoint (config.json + model.safetensors with the transformers Qwen2-VL parameter names) loads
    into the native weight layout and produces the same logits as weights passed directly."""
    import json
    from safetensors.torch import save_file
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration
    from quickvideo_amd.lvu import load_native_model
    cfg = Qwen2VLConfig(
        text_config=dict(hidden_size=256, num_attention_heads=2, num_key_value_heads=1, intermediate_size=512, num_hidden_layers=2,
                         vocab_size=320, rms_norm_eps=1e-6, tie_word_embeddings=False,
                         rope_parameters=dict(rope_type="default", mrope_section=[16, 24, 24], rope_theta=1_000_000.0)),
        vision_config=dict(depth=2, embed_dim=64, hidden_size=256, num_heads=4, mlp_ratio=2, patch_size=14, spatial_merge_size=2,
                           temporal_patch_size=2),
        video_token_id=300, vision_start_token_id=301, vision_end_token_id=302)
    torch.manual_seed(0)
    hf = Qwen2VLForConditionalGeneration(cfg).eval()
    sd = {k: v.contiguous() for k, v in hf.state_dict().items()}
    save_file(sd, str(tmp_path / "model.safetensors"))
    d = cfg.to_dict()
    d["text_config"]["rope_scaling"] = {"mrope_section": [16, 24, 24]}
    d["text_config"]["rope_theta"] = 1_000_000.0
    json.dump(d, open(tmp_path / "config.json", "w"), default=str)
    m = load_native_model(str(tmp_path), device="cpu")
    s = m.spec
    assert (s.hidden, s.n_heads, s.n_kv_heads, s.head_dim, s.intermediate, s.n_layers, s.vocab) == (256, 2, 1, 128, 512, 2, 320)
    assert (s.video_token_id, s.vision_start_token_id) == (300, 301) and tuple(s.mrope_section) == (16, 24, 24)
    lm = hf.model.language_model
    assert torch.equal(m.text.layers[1].w_qkv[:256].float(), lm.layers[1].self_attn.q_proj.weight.to(torch.bfloat16).float())
    assert torch.equal(m.text.layers[0].w_gate_up[512:].float(), lm.layers[0].mlp.up_proj.weight.to(torch.bfloat16).float())
    assert torch.equal(m.text.lm_head.float(), hf.lm_head.weight.to(torch.bfloat16).float())
    assert m.vision.spec.depth == 2 and m.vision.spec.embed_dim == 64 and m.vision.spec.out_hidden == 256
    assert torch.equal(m.vision.blocks[1].fc1_w.float(), hf.model.visual.blocks[1].mlp.fc1.weight.to(torch.bfloat16).float())
    # and it runs end to end through the drop-in API
    import lvu
    obj = lvu.LVU(lvu.LVUConfig(str(tmp_path), top_p=0.5, video_group_size=4, num_frames=8), model=m)
    obj._ops = OracleOps()
    out = obj.generate("hi", "synthetic://?frames=16&h=56&w=84&seed=1", max_new_tokens=2)
    assert obj.generate("hi", "synthetic://?frames=16&h=56&w=84&seed=1", max_new_tokens=2, num_beams=1) == out
    assert obj.generate("hi", "synthetic://?frames=16&h=56&w=84&seed=1", max_new_tokens=2, num_beams=2)[0].count("<tok_") == 2   # beam search (round 4)
    assert obj.generate("hi", "synthetic://?frames=16&h=56&w=84&seed=1", max_new_tokens=2, num_beams=2, do_sample=True,
                        seed=3)[0].count("<tok_") >= 1               # beam-search multinomial sampling
    assert obj.generate("hi", "synthetic://?frames=16&h=56&w=84&seed=1", max_new_tokens=2, do_sample=True, temperature=0.7,
                        seed=1)[0].count("<tok_") == 2
    assert obj.generate("hi", "synthetic://?frames=16&h=56&w=84&seed=1", max_new_tokens=2, do_sample=True, top_k=1) == out
    assert out[0].count("<tok_") == 2


# ---------------------------------------------------------------- group-token parallel ("sp") over gloo
def _sp_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from quickvideo_amd.engine import QuickPrefillEngine
    from quickvideo_amd.lvu_config import LVUConfig
    from quickvideo_amd.spec import TextSpec
    from quickvideo_amd.weights import DecoderWeights
    so = O.TextSpec(hidden=256, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=256, n_layers=2, vocab=128)
    spec = TextSpec(hidden=256, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=256, n_layers=2, vocab=128)
    w = {k: v.to(torch.bfloat16) for k, v in O.synthetic_text_weights(so, seed=5, norm_jitter=0.1).items()}
    rs = np.random.RandomState(9)
    groups = [64 * world + 5, 64 * world + 70]          # uneven split: the last rank gets fewer tokens
    T = sum(groups) + 9
    embeds = torch.from_numpy(rs.standard_normal((T, 256)).astype(np.float32) * 0.5).to(torch.bfloat16)
    pos = np.tile(np.arange(T, dtype=np.int64), (3, 1))
    cfg = LVUConfig("x", top_p=0.5, video_group_size=4)
    eng = QuickPrefillEngine(DecoderWeights.from_named(spec, w, "cpu"), cfg, capacity=T + 4, max_group_tokens=max(groups), device="cpu",
                             ops=OracleOps(), sp_group=dist.group.WORLD, sp_rank=rank, sp_size=world)
    eng.kept_trace = []
    post = torch.from_numpy(pos)
    st = 0
    for n in groups:
        eng.prefill_group(embeds[st:st + n], post[:, st:st + n]); st += n
    logits = eng.prefill_tail(embeds[st:], post[:, st:])          # 9 tokens < 64*world: replicated path
    kept = [None if k is None else k.numpy().copy() for _, k in eng.kept_trace]
    if rank == 0:
        ref = O.group_prefill(w, so, embeds, pos, groups, O.PruneCfg(top_p=0.5))
        ret["ref_logits"], ret["ref_len"] = ref["logits"].numpy(), ref["cache_len"]
        ret["ref_kept"] = [k for g in ref["kept"] for k in g]
        ret["ref_k0"] = ref["cache"].k[0].float().numpy()
    ret[f"logits{rank}"], ret[f"kept{rank}"], ret[f"len{rank}"] = logits.numpy(), kept, list(eng.arena.len)
    ret[f"k0_{rank}"] = eng.arena.k(0)[:, :eng.arena.len[0]].float().numpy()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_group_token_parallel_gloo(world):
    port = 33500 + os.getpid() % 2000 + world
    ret = mp.Manager().dict()
    mp.spawn(_sp_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[f"len{r}"] == ret["ref_len"]
        assert np.array_equal(ret[f"k0_{r}"], ret["k0_0"])                       # every rank holds the identical arena replica
        assert np.array_equal(ret[f"logits{r}"], ret["logits0"])
        for a, b in zip(ret[f"kept{r}"], ret["kept0"]):
            assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))
    tot = same = 0
    for a, rk in zip(ret["kept0"], ret["ref_kept"]):
        if rk is not None:
            tot += len(rk); same += len(set(a.tolist()) & set(rk.tolist()))
    assert same / tot >= 0.95
    assert np.max(np.abs(ret["logits0"] - ret["ref_logits"])) <= 4e-2
    assert np.max(np.abs(ret["k0_0"] - ret["ref_k0"])) <= 0.25                   # layer-0 keys: per-row GEMM rounding only


# ---------------------------------------------------------------- layer pipeline over gloo (world_size 2 and 3)
def _pp_worker(rank, world, port, ret, pps=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from quickvideo_amd.engine import QuickPrefillEngine
    from quickvideo_amd.lvu_config import LVUConfig
    from quickvideo_amd.spec import TextSpec
    from quickvideo_amd.weights import DecoderWeights, pp_layer_split
    L = 4
    so = O.TextSpec(hidden=256, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=256, n_layers=L, vocab=128)
    spec = TextSpec(hidden=256, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=256, n_layers=L, vocab=128)
    w = {k: v.to(torch.bfloat16) for k, v in O.synthetic_text_weights(so, seed=7, norm_jitter=0.1).items()}
    rs = np.random.RandomState(3)
    groups = [37, 50, 41]
    T = sum(groups) + 7
    embeds = torch.from_numpy(rs.standard_normal((T, 256)).astype(np.float32) * 0.5).to(torch.bfloat16)
    pos = torch.from_numpy(np.tile(np.arange(T, dtype=np.int64), (3, 1)))
    cfg = LVUConfig("x", top_p=0.5, video_group_size=4, top_k_decay_type="linear", top_k_decay_factor=0.5,   # decay: global layer index matters
                    prefill_prune_starting_layer=pps)                   # pps: the hidden rows shrink from that layer on, across stages

    def run(eng):
        eng.kept_trace = []
        st = 0
        for n in groups:
            eng.prefill_group(embeds[st:st + n], pos[:, st:st + n]); st += n
        logits = eng.prefill_tail(embeds[st:], pos[:, st:])
        tok_in = torch.from_numpy(rs.standard_normal((1, 256)).astype(np.float32)).to(torch.bfloat16)
        logits2 = eng.decode_step(tok_in, rope_delta=0)
        return logits, logits2

    l0, l1 = pp_layer_split(L, world, rank)
    stage = QuickPrefillEngine(DecoderWeights.from_named(spec, w, "cpu", layer_range=(l0, l1)), cfg, capacity=T + 8, max_group_tokens=max(groups),
                               device="cpu", ops=OracleOps(), pp_group=dist.group.WORLD, pp_rank=rank, pp_size=world)
    logits, logits2 = run(stage)
    ret[f"len{rank}"] = list(stage.arena.len)
    ret[f"kept{rank}"] = [(l, None if k is None else k.numpy().copy()) for l, k in stage.kept_trace]
    if rank == world - 1:
        ret["logits"], ret["logits2"] = logits.numpy(), logits2.numpy()
    else:
        assert logits is None and logits2 is None
    if rank == 0:                                             # the same model in one process
        rs = np.random.RandomState(3); rs.standard_normal((T, 256))    # replay the decode embedding draw
        full = QuickPrefillEngine(DecoderWeights.from_named(spec, w, "cpu"), cfg, capacity=T + 8, max_group_tokens=max(groups), device="cpu",
                                  ops=OracleOps())
        a, b = run(full)
        ret["ref_logits"], ret["ref_logits2"], ret["ref_len"] = a.numpy(), b.numpy(), list(full.arena.len)
        ret["ref_kept"] = [(l, None if k is None else k.numpy().copy()) for l, k in full.kept_trace]
    dist.destroy_process_group()


@pytest.mark.parametrize("world,pps", [(2, None), (3, None), (2, 1), (3, 0), (4, 2)])
def test_layer_pipeline_gloo(world, pps):
    """Layer-pipeline stages reproduce the single-process engine exactly (same ops in the same order; the hand-off is a copy) — also
    with hidden-state pruning (prefill_prune_starting_layer): the hand-off then carries the surviving rows and their original indices."""
    port = 35500 + os.getpid() % 2000 + world * 7 + (0 if pps is None else pps + 1)
    ret = mp.Manager().dict()
    mp.spawn(_pp_worker, args=(world, port, ret, pps), nprocs=world, join=True)
    from quickvideo_amd.weights import pp_layer_split
    assert np.array_equal(ret["logits"], ret["ref_logits"]) and np.array_equal(ret["logits2"], ret["ref_logits2"])
    lens = []
    for r in range(world):
        lens += ret[f"len{r}"]
    assert lens == ret["ref_len"]
    # kept indices per (segment, layer): stage r's local layer l is global layer l0 + l
    ref = ret["ref_kept"]
    L = 4
    n_seg = len(ref) // L
    for r in range(world):
        l0, l1 = pp_layer_split(L, world, r)
        mine = ret[f"kept{r}"]
        assert len(mine) == n_seg * (l1 - l0)
        for s in range(n_seg):
            for l in range(l1 - l0):
                a, b = mine[s * (l1 - l0) + l][1], ref[s * L + l0 + l][1]
                assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))


def _pp_query_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from quickvideo_amd.engine import QuickPrefillEngine
    from quickvideo_amd.lvu_config import LVUConfig
    from quickvideo_amd.spec import TextSpec
    from quickvideo_amd.weights import DecoderWeights, pp_layer_split
    L = 4
    dims = dict(hidden=256, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=256, n_layers=L, vocab=128)
    so, spec = O.TextSpec(**dims), TextSpec(**dims)
    w = {k: v.to(torch.bfloat16) for k, v in O.synthetic_text_weights(so, seed=7, norm_jitter=0.1).items()}
    rs = np.random.RandomState(5)
    groups, m = [33, 40, 29], 9
    T = sum(groups) + m
    embeds = torch.from_numpy(rs.standard_normal((T, 256)).astype(np.float32) * 0.5).to(torch.bfloat16)
    pos = torch.from_numpy(np.tile(np.arange(T, dtype=np.int64), (3, 1)))
    cfg = LVUConfig("x", top_p=0.5, video_group_size=4, top_k_predict_type="query_attention_weights")

    def run(eng):
        eng.kept_trace = []
        st = 0
        for n in groups:
            eng.prefill_group(embeds[st:st + n], pos[:, st:st + n + m], prompt_embeds=embeds[-m:]); st += n
        return eng.prefill_tail(embeds[st:], pos[:, st:])

    l0, l1 = pp_layer_split(L, world, rank)
    stage = QuickPrefillEngine(DecoderWeights.from_named(spec, w, "cpu", layer_range=(l0, l1)), cfg, capacity=T + 8, max_group_tokens=max(groups) + m,
                               device="cpu", ops=OracleOps(), pp_group=dist.group.WORLD, pp_rank=rank, pp_size=world)
    logits = run(stage)
    ret[f"len{rank}"] = list(stage.arena.len)
    ret[f"kept{rank}"] = [None if k is None else k.numpy().copy() for _, k in stage.kept_trace]
    if rank == world - 1:
        ret["logits"] = logits.numpy()
    if rank == 0:
        full = QuickPrefillEngine(DecoderWeights.from_named(spec, w, "cpu"), cfg, capacity=T + 8, max_group_tokens=max(groups) + m, device="cpu",
                                  ops=OracleOps())
        ret["ref_logits"], ret["ref_len"] = run(full).numpy(), list(full.arena.len)
        ret["ref_kept"] = [None if k is None else k.numpy().copy() for _, k in full.kept_trace]
        ref = O.group_prefill(w, so, embeds, pos.numpy(), groups, O.PruneCfg(top_p=0.5, top_k_predict_type="query_attention_weights"))
        ret["oracle_len"], ret["oracle_logits"] = ref["cache_len"], ref["logits"].numpy()
    dist.destroy_process_group()


def test_layer_pipeline_with_query_score_pruning_gloo():
    """Query-attention-score pruning (prompt-appended groups, lvu_cache.py:97-117) across two layer-pipeline stages: the n + m rows travel
    between the stages; cache lengths, kept lists and logits equal the single-process engine's, which equals the oracle."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_pp_query_worker, args=(world, 36900 + os.getpid() % 2000, ret), nprocs=world, join=True)
    assert ret["len0"] + ret["len1"] == ret["ref_len"] == ret["oracle_len"]
    assert np.array_equal(ret["logits"], ret["ref_logits"]) and np.array_equal(ret["logits"], ret["oracle_logits"])
    L, segs = 4, 4
    for s in range(segs):
        for l in range(L):
            a = ret["kept0"][s * 2 + l] if l < 2 else ret["kept1"][s * 2 + l - 2]
            b = ret["ref_kept"][s * L + l]
            assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))


def _tp_query_worker(rank, world, port, ret, pt):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from quickvideo_amd.engine import QuickPrefillEngine
    from quickvideo_amd.lvu_config import LVUConfig
    from quickvideo_amd.spec import TextSpec
    from quickvideo_amd.weights import DecoderWeights
    dims = dict(hidden=256, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=256, n_layers=2, vocab=128)
    so, spec = O.TextSpec(**dims), TextSpec(**dims)
    w = {k: v.to(torch.bfloat16) for k, v in O.synthetic_text_weights(so, seed=9, norm_jitter=0.1).items()}
    rs = np.random.RandomState(6)
    groups, m = [33, 40], 9
    T = sum(groups) + m
    embeds = torch.from_numpy(rs.standard_normal((T, 256)).astype(np.float32) * 0.5).to(torch.bfloat16)
    pos = torch.from_numpy(np.tile(np.arange(T, dtype=np.int64), (3, 1)))
    cfg = LVUConfig("x", top_p=0.5, video_group_size=4, top_k_predict_type=pt)

    def run(eng):
        eng.kept_trace = []
        st = 0
        for n in groups:
            eng.prefill_group(embeds[st:st + n], pos[:, st:st + n + m], prompt_embeds=embeds[-m:]); st += n
        return eng.prefill_tail(embeds[st:], pos[:, st:])

    eng = QuickPrefillEngine(DecoderWeights.from_named(spec, w, "cpu", tp_rank=rank, tp_size=world), cfg, capacity=T + 8, max_group_tokens=max(groups) + m,
                             device="cpu", ops=OracleOps(), tp_group=dist.group.WORLD)
    logits = run(eng)
    ret[f"len{rank}"], ret[f"logits{rank}"] = list(eng.arena.len), logits.numpy()
    ret[f"kept{rank}"] = [None if k is None else k.numpy().copy() for _, k in eng.kept_trace]
    if rank == 0:
        full = QuickPrefillEngine(DecoderWeights.from_named(spec, w, "cpu"), cfg, capacity=T + 8, max_group_tokens=max(groups) + m, device="cpu", ops=OracleOps())
        ret["ref_logits"], ret["ref_len"] = run(full).numpy(), list(full.arena.len)
        ret["ref_kept"] = [None if k is None else k.numpy().copy() for _, k in full.kept_trace]
    dist.destroy_process_group()


@pytest.mark.parametrize("pt", ["query_attention_weights", "query_attention_weights_by_value_norm"])
def test_query_score_pruning_under_tensor_parallelism_gloo(pt):
    """VERDICT r3 Missing #5 (f4 x TP): heads sharded over two ranks; every rank sums its heads' probabilities, the per-head sums are
    all-gathered in head order and every rank takes the same mean over all heads (qp_query_head_sums + qp_query_scores_from_head_sums; the
    single-device scoring IS that composition).  Layer 0 sees identical inputs: its kept list must equal the single-process engine's exactly;
    both ranks agree on every list; cache lengths equal; logits to bf16 all-reduce noise."""
    ret = mp.Manager().dict()
    mp.spawn(_tp_query_worker, args=(2, 37300 + os.getpid() % 2000, ret, pt), nprocs=2, join=True)
    assert ret["len0"] == ret["len1"] == ret["ref_len"]
    for a, b in zip(ret["kept0"], ret["kept1"]):
        assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))
    assert np.array_equal(ret["kept0"][0], ret["ref_kept"][0])                       # (group 0, layer 0)
    tot = sum(len(a) for a in ret["ref_kept"] if a is not None)
    same = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(ret["kept0"], ret["ref_kept"]) if a is not None)
    assert same / tot >= 0.9, same / tot
    assert np.array_equal(ret["logits0"], ret["logits1"]) and np.max(np.abs(ret["logits0"] - ret["ref_logits"])) <= 6e-2


def _pp_order_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    from quickvideo_amd.engine import QuickPrefillEngine
    from quickvideo_amd.lvu_config import LVUConfig
    from quickvideo_amd.spec import TextSpec
    from quickvideo_amd.weights import DecoderWeights, pp_layer_split
    L = 2
    so = O.TextSpec(hidden=256, n_heads=2, n_kv_heads=1, head_dim=128, intermediate=256, n_layers=L, vocab=64)
    spec = TextSpec(hidden=256, n_heads=2, n_kv_heads=1, head_dim=128, intermediate=256, n_layers=L, vocab=64)
    w = {k: v.to(torch.bfloat16) for k, v in O.synthetic_text_weights(so, seed=3).items()}
    eng = QuickPrefillEngine(DecoderWeights.from_named(spec, w, "cpu", layer_range=pp_layer_split(L, 2, rank)), LVUConfig("x", top_p=0.5, video_group_size=4),
                             capacity=64, max_group_tokens=16, device="cpu", ops=OracleOps(), pp_rank=rank, pp_size=2)
    segs = [torch.full((9 + i, 256), float(i + 1), dtype=torch.bfloat16) for i in range(4)]
    if rank == 0:
        t = []
        for i, h in enumerate(segs):
            t0 = time.perf_counter()
            hh = h.clone()
            eng._seg_rows = None
            eng._pp_out(hh)
            hh.zero_()                                        # the caller's buffer is reused at once (b_h is): the slot must hold a COPY
            t.append(time.perf_counter() - t0)
        eng.pp_flush()
        ret["send_s"] = t
    else:
        time.sleep(1.0)                                       # the receiver is late: the first TWO hand-offs must not wait for it
        got = []
        for h in segs:
            x, rows = eng._pp_in(torch.empty(0), h.shape[0], prune=False)
            got.append(x.clone())
            time.sleep(0.05)
        ret["ok"] = all(torch.equal(a, b) for a, b in zip(got, segs))
    dist.barrier()
    dist.destroy_process_group()


def test_layer_pipeline_handoff_is_asynchronous_ordered_and_double_buffered():
    """VERDICT r3 #5c: the stage-to-stage hand-off as RCCL will run it — `isend` from one of two send slots, no host staging for CPU/RCCL
    tensors, a blocking wait only when a slot is reused.  A receiver that shows up a second late: the sender's first two hand-offs return
    immediately (its compute would go on), the third waits for slot 0 to drain; all four segments (different row counts) arrive in order
    with the bytes they had when handed off, although the sender overwrote its source buffer right after each call."""
    ret = mp.Manager().dict()
    mp.spawn(_pp_order_worker, args=(2, 33100 + os.getpid() % 2000, ret), nprocs=2, join=True)
    assert ret["ok"]
    s0, s1, s2, s3 = ret["send_s"]
    assert s0 < 0.3 and s1 < 0.3, ret["send_s"]               # not held back by the late receiver
    assert s2 > 0.5, ret["send_s"]                            # slot 0 is reused only after its send has completed


# ---------------------------------------------------------------- layer pipeline of group-token parallel stages (pp2 x sp2)
def _ppsp_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from quickvideo_amd.engine import QuickPrefillEngine
    from quickvideo_amd.lvu_config import LVUConfig
    from quickvideo_amd.spec import TextSpec
    from quickvideo_amd.weights import DecoderWeights, pp_layer_split
    pp, sp = 2, 2
    stage, sp_rank = rank // sp, rank % sp
    groups_ = [dist.new_group(ranks=list(range(s * sp, (s + 1) * sp))) for s in range(pp)]     # every rank creates every group
    L = 4
    so = O.TextSpec(hidden=256, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=256, n_layers=L, vocab=128)
    spec = TextSpec(hidden=256, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=256, n_layers=L, vocab=128)
    w = {k: v.to(torch.bfloat16) for k, v in O.synthetic_text_weights(so, seed=11, norm_jitter=0.1).items()}
    rs = np.random.RandomState(4)
    groups = [64 * sp + 9, 64 * sp + 40]
    T = sum(groups) + 8
    embeds = torch.from_numpy(rs.standard_normal((T, 256)).astype(np.float32) * 0.5).to(torch.bfloat16)
    pos = np.tile(np.arange(T, dtype=np.int64), (3, 1))
    post = torch.from_numpy(pos)
    cfg = LVUConfig("x", top_p=0.5, video_group_size=4)
    eng = QuickPrefillEngine(DecoderWeights.from_named(spec, w, "cpu", layer_range=pp_layer_split(L, pp, stage)), cfg, capacity=T + 4,
                             max_group_tokens=max(groups), device="cpu", ops=OracleOps(), sp_group=groups_[stage], sp_rank=sp_rank, sp_size=sp,
                             pp_rank=stage, pp_size=pp, pp_peers=[s * sp + sp_rank for s in range(pp)])
    st = 0
    for n in groups:
        eng.prefill_group(embeds[st:st + n], post[:, st:st + n]); st += n
    logits = eng.prefill_tail(embeds[st:], post[:, st:])
    ret[f"len{rank}"] = list(eng.arena.len)
    ret[f"logits{rank}"] = None if logits is None else logits.numpy()
    if rank == 0:
        ref = O.group_prefill(w, so, embeds, pos, groups, O.PruneCfg(top_p=0.5))
        ret["ref_logits"], ret["ref_len"] = ref["logits"].numpy(), ref["cache_len"]
    dist.destroy_process_group()


def test_layer_pipeline_of_group_token_parallel_stages_gloo():
    world = 4
    port = 36500 + os.getpid() % 2000
    ret = mp.Manager().dict()
    mp.spawn(_ppsp_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret["logits0"] is None and ret["logits1"] is None                    # stage 0 has no logits
    assert np.array_equal(ret["logits2"], ret["logits3"])                        # both sp ranks of the last stage agree exactly
    assert np.max(np.abs(ret["logits2"] - ret["ref_logits"])) <= 4e-2
    assert ret["len0"] + ret["len2"] == ret["ref_len"] and ret["len1"] == ret["len0"] and ret["len3"] == ret["len2"]


def test_duck_typed_reader_and_eos_default():
    """A third-party reader with the InterleavedVideoReader contract (no VideoReaderBase inheritance) is accepted; an object without
    it is refused with a TypeError naming the contract; chat() stops at the processor's end-of-turn token by default."""
    import lvu
    from quickvideo_amd.frames import open_video
    from quickvideo_amd.lvu import load_native_model

    class ThirdPartyReader:                       # the five members, nothing else
        height = width = None
        interpolation = "LANCZOS"
        frame_iter = 4

        def __init__(self, n=40, h=56, w=84):
            self.n, self.src_h, self.src_w, self._idx, self._cur = n, h, w, None, 0

        def __len__(self): return self.n
        def get_fps(self): return 2.0
        def process(self, idx): self._idx, self._cur = list(idx), 0

        def __next__(self):
            if self._cur >= len(self._idx):
                raise StopIteration
            sel = self._idx[self._cur:self._cur + self.frame_iter]
            self._cur += len(sel)
            rs = np.random.RandomState(7)
            return torch.from_numpy(np.stack([rs.randint(0, 256, (3, self.height, self.width), dtype=np.uint8) for _ in sel]))

    r = ThirdPartyReader()
    assert open_video(r) is r
    with pytest.raises(TypeError):
        open_video(object())
    m = load_native_model("synthetic:tiny", device="cpu")
    obj = lvu.LVU(lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=4, num_frames=8), model=m)
    obj._ops = OracleOps()
    out = obj.generate("What happens?", ThirdPartyReader(), max_new_tokens=3)
    assert len(out) == 1 and 1 <= out[0].count("<tok_") <= 3
    # EOS default: make the first generated token the end-of-turn token -> decoding stops after it
    first = int(out[0].split("<tok_")[1].split(">")[0])
    obj.processor.eos_token_id = first            # (processor.tokenizer is the processor itself for the offline stand-in)
    out2 = obj.generate("What happens?", ThirdPartyReader(), max_new_tokens=3)
    assert out2[0].count("<tok_") == 1
    assert obj.generate("What happens?", ThirdPartyReader(), max_new_tokens=3, eos_token_id=None)[0].count("<tok_") == 3


def test_load_qwen25_vl_checkpoint_directory(tmp_path):
    """A Qwen2.5-VL checkpoint directory (the reference's model family, lvu.py:60): vision_config carries hidden_size /
    out_hidden_size / intermediate_size / fullatt_block_indexes instead of embed_dim; the temporal M-RoPE scale follows the sampled
    fps (tokens_per_second * temporal_patch_size / fps) instead of a hard-coded 2.0."""
    import json
    from safetensors.torch import save_file
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLConfig
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VLForConditionalGeneration
    from quickvideo_amd.frames import open_video
    from quickvideo_amd.lvu import load_native_model
    from quickvideo_amd.pipeline import PrefillPipeline
    from quickvideo_amd.processor import SyntheticProcessor
    import lvu
    cfg = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=256, num_attention_heads=2, num_key_value_heads=1, intermediate_size=512, num_hidden_layers=2,
                         vocab_size=320, rms_norm_eps=1e-6, tie_word_embeddings=False,
                         rope_parameters=dict(rope_type="default", mrope_section=[16, 24, 24], rope_theta=1_000_000.0)),
        vision_config=dict(depth=2, hidden_size=64, intermediate_size=80, num_heads=4, out_hidden_size=256, window_size=112,
                           fullatt_block_indexes=[1], patch_size=14, spatial_merge_size=2, temporal_patch_size=2, tokens_per_second=2),
        video_token_id=300, vision_start_token_id=301, vision_end_token_id=302)
    torch.manual_seed(0)
    hf = Qwen2_5_VLForConditionalGeneration(cfg).eval()
    save_file({k: v.contiguous() for k, v in hf.state_dict().items()}, str(tmp_path / "model.safetensors"))
    d = cfg.to_dict()
    d["text_config"]["rope_scaling"] = {"mrope_section": [16, 24, 24]}
    d["text_config"]["rope_theta"] = 1_000_000.0
    json.dump(d, open(tmp_path / "config.json", "w"), default=str)
    m = load_native_model(str(tmp_path), device="cpu")
    vs = m.vision.spec
    assert (vs.arch, vs.depth, vs.embed_dim, vs.intermediate, vs.out_hidden, vs.fullatt_blocks) == ("qwen2.5", 2, 64, 80, 256, (1,))
    assert m.spec.temporal_scale == -2.0                                # "derive from the sampled fps" marker
    pipe = PrefillPipeline(m, lvu.LVUConfig("x", top_p=0.5, video_group_size=4, num_frames=8), SyntheticProcessor(m.spec), ops=OracleOps())
    # 40 frames at 2 fps = 20 s, 8 frames sampled -> 0.4 fps -> second_per_grid = 2 / 0.4 = 5 s -> 10 temporal ids per grid step
    P = pipe.plan(open_video("synthetic://?frames=40&h=56&w=84&fps=2"), "q")
    t_ids = P["pos"][0, len(P["prompt"].prefix_ids):len(P["prompt"].prefix_ids) + (P["nframes"] // 2) * (P["gh"] // 2) * (P["gw"] // 2)]
    per_grid = (P["gh"] // 2) * (P["gw"] // 2)
    assert int(t_ids[per_grid] - t_ids[0]) == 10


@pytest.mark.parametrize("family,proc", [("qwen2.5-vl", "synthetic"), ("qwen2-vl", "synthetic"), ("qwen2-vl", "hf")])
def test_qwen_vl_end_to_end_equals_hf_forward(tmp_path, family, proc):
    """The reference's model family end to end (lvu.py:60; and Qwen2-VL, BASELINE.json's model): a tiny checkpoint through LVU's pipeline — frames -> patchify
    -> windowed ViT + merger -> group-chunked prefill (2 groups, rho = 1: no pruning) -> prompt tail — against ONE forward of the
    installed transformers Qwen2_5_VLForConditionalGeneration over the whole prompt (it derives its own M-RoPE index from
    video_grid_thw and second_per_grid_ts).  Same bf16-representable weights; engine in bf16, HF in fp32: stated tolerance
    |dlogit| <= 5e-2 (|logit| <= 1), cosine >= 0.999, same first token."""
    import json
    from safetensors.torch import save_file
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLConfig
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VLForConditionalGeneration
    from quickvideo_amd.engine import QuickPrefillEngine
    from quickvideo_amd.frames import open_video
    from quickvideo_amd.lvu import load_native_model
    from quickvideo_amd.pipeline import PrefillPipeline
    from quickvideo_amd.processor import SyntheticProcessor
    from quickvideo_amd.vit import patchify_frames
    import lvu
    text = dict(hidden_size=256, num_attention_heads=2, num_key_value_heads=1, intermediate_size=512, num_hidden_layers=2,
                vocab_size=320, rms_norm_eps=1e-6, tie_word_embeddings=False,
                rope_parameters=dict(rope_type="default", mrope_section=[16, 24, 24], rope_theta=1_000_000.0))
    ids_kw = dict(video_token_id=300, vision_start_token_id=301, vision_end_token_id=302)
    if family == "qwen2.5-vl":
        cfg = Qwen2_5_VLConfig(text_config=text, **ids_kw,
                               vision_config=dict(depth=2, hidden_size=64, intermediate_size=80, num_heads=4, out_hidden_size=256, window_size=112,
                                                  fullatt_block_indexes=[1], patch_size=14, spatial_merge_size=2, temporal_patch_size=2,
                                                  tokens_per_second=2))
        cls = Qwen2_5_VLForConditionalGeneration
    else:
        from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration
        cfg = Qwen2VLConfig(text_config=text, **ids_kw,
                            vision_config=dict(depth=2, embed_dim=64, hidden_size=256, num_heads=4, mlp_ratio=2, patch_size=14,
                                               spatial_merge_size=2, temporal_patch_size=2))
        cls = Qwen2VLForConditionalGeneration
    cfg._attn_implementation = "eager"
    torch.manual_seed(0)
    hf = cls(cfg).eval()
    with torch.no_grad():
        for p in hf.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    save_file({k: v.contiguous() for k, v in hf.state_dict().items()}, str(tmp_path / "model.safetensors"))
    d = cfg.to_dict()
    d["text_config"]["rope_scaling"] = {"mrope_section": [16, 24, 24]}
    d["text_config"]["rope_theta"] = 1_000_000.0
    json.dump(d, open(tmp_path / "config.json", "w"), default=str)
    m = load_native_model(str(tmp_path), device="cpu")
    frames = torch.from_numpy(np.random.RandomState(5).randint(0, 256, (8, 3, 112, 168), dtype=np.uint8))
    video = str(tmp_path / "v.npy")
    np.save(video, frames.numpy())
    if proc == "hf":            # the processor seam: the INSTALLED transformers Qwen2VLProcessor object (tiny tokenizer, Qwen2-VL chat template)
        from tests.test_processor_seam import installed_qwen2vl_processor
        processor = installed_qwen2vl_processor((4, 8, 12))
    else:
        processor = SyntheticProcessor(m.spec)
    pipe = PrefillPipeline(m, lvu.LVUConfig("x", top_p=1.0, video_group_size=4, num_frames=8), processor, ops=OracleOps())
    cap, orig = {}, QuickPrefillEngine.prefill_tail

    def tail(self, e, p):
        cap["logits"] = orig(self, e, p)
        return cap["logits"]
    QuickPrefillEngine.prefill_tail = tail
    try:
        toks = pipe.generate("what is this", video, max_new_tokens=1, overlap=False)
    finally:
        QuickPrefillEngine.prefill_tail = orig
    rd = open_video(video)
    P = pipe.plan(rd, "what is this")
    assert len(P["plan"].tokens) == 2                                    # group-chunked: the second group attends over the first
    n_video = (P["nframes"] // 2) * (P["gh"] // 2) * (P["gw"] // 2)
    ids = torch.tensor([list(P["prompt"].prefix_ids) + [300] * n_video + list(P["prompt"].tail_ids)])
    rows, grid = patchify_frames(frames[torch.from_numpy(P["idx"])], m.vision.spec, torch.float32)
    sample_fps = P["nframes"] / (len(rd) / rd.get_fps())
    extra = dict(mm_token_type_ids=(ids == 300).long() * 2)               # transformers 5.x: text 0 / image 1 / video 2 (both families)
    if family == "qwen2.5-vl":
        extra["second_per_grid_ts"] = torch.tensor([2.0 / sample_fps])
    with torch.no_grad():
        out = hf(input_ids=ids, pixel_values_videos=rows, video_grid_thw=torch.tensor([list(grid)]), attention_mask=torch.ones_like(ids), **extra)
    ref, got = out.logits[0, -1].float(), cap["logits"].float()
    assert (ref - got).abs().max().item() <= 5e-2, (ref - got).abs().max().item()
    assert torch.nn.functional.cosine_similarity(ref, got, 0).item() >= 0.999
    assert int(ref.argmax()) == int(got.argmax()) == toks[0]


def test_producer_ring_fill_is_native_and_gil_free():
    """SURVEY 8b threading row: the producer's copy into the pinned ring must not starve the thread that launches the kernels.
    qp_host_memcpy through ctypes (the GIL is dropped around the foreign call): correct for odd sizes, and the main thread keeps
    its Python iteration rate while a second thread copies 50 MB blocks back to back."""
    import threading
    import time
    from quickvideo_amd.native import host_memcpy
    a = torch.randint(0, 255, (8, 3, 1080, 1920), dtype=torch.uint8)
    b = torch.empty_like(a)
    host_memcpy(b, a)
    assert torch.equal(a, b)
    c, d = torch.empty(5, dtype=torch.uint8), torch.arange(5, dtype=torch.uint8)
    host_memcpy(c, d, 8)
    assert torch.equal(c, d)

    def rate(worker, dur=0.5):
        stop = [False]

        def loop():
            while not stop[0] and worker is not None:
                worker()
        th = threading.Thread(target=loop)
        th.start()
        time.sleep(0.05)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < dur:
            n += 1
        stop[0] = True
        th.join()
        return n

    alone = rate(None)
    native = rate(lambda: host_memcpy(b, a, 2))
    assert native > 0.3 * alone, (native, alone)        # a GIL-holding 10 ms copy loop would leave ~1/3 or less (5 ms switch interval)


def test_generation_kwargs_like_hf_generate():
    """The reference hands **generation_kwargs to HF `generate` (qwen25_lvu.py:744-761).  TokenSelector restates its logits processing:
    (a) processed scores equal the installed transformers processors / warpers on random logits, (b) through LVU.generate: sampling
    is reproducible under a seed, top_k=1 sampling equals greedy, a large repetition penalty changes the greedy continuation,
    beam search is refused, generation_config.json defaults are honoured."""
    import lvu
    from transformers.generation.logits_process import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper,
                                                        TopPLogitsWarper)
    from quickvideo_amd.lvu import load_native_model
    from quickvideo_amd.sampling import TokenSelector
    g = torch.Generator().manual_seed(0)
    V = 5000
    for trial, (T, k, p, rp) in enumerate([(0.7, 50, 0.9, 1.05), (1.0, 0, 0.5, 1.3), (1e-6, 1, 0.001, 1.05), (2.0, 7, 1.0, 1.0), (0.3, 0, 0.05, 1.0)] * 3):
        logits = torch.randn(V, generator=g) * 3
        ids = torch.randint(0, V, (1, 50), generator=g)
        ref = logits[None].clone()
        if rp != 1.0: ref = RepetitionPenaltyLogitsProcessor(rp)(ids, ref)
        if T != 1.0: ref = TemperatureLogitsWarper(T)(ids, ref)
        if k > 0: ref = TopKLogitsWarper(k)(ids, ref)
        if p < 1.0: ref = TopPLogitsWarper(p)(ids, ref)
        ts = TokenSelector(True, T, k, p, rp, seed=1)
        ts.observe(ids[0].tolist(), V, "cpu")
        got = ts.process(logits)
        assert torch.equal(torch.isinf(got), torch.isinf(ref[0])) and torch.allclose(got[~torch.isinf(got)], ref[0][~torch.isinf(ref[0])])
    with pytest.raises(ValueError):
        TokenSelector(True, temperature=0.0)
    m = load_native_model("synthetic:tiny", device="cpu")
    obj = lvu.LVU(lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=4, num_frames=8), model=m)
    obj._ops = OracleOps()
    video = "synthetic://?frames=16&h=56&w=84&fps=2&seed=3"
    run = lambda **kw: obj.generate("What happens?", video, max_new_tokens=6, eos_token_id=None, **kw)[0]
    greedy = run()
    assert run(do_sample=True, top_k=1) == greedy                                        # Qwen2-VL's shipped config without the penalty
    a, b = run(do_sample=True, temperature=5.0, seed=11), run(do_sample=True, temperature=5.0, seed=11)
    assert a == b and a.count("<tok_") == 6
    assert len({run(do_sample=True, temperature=5.0, seed=s) for s in range(4)}) > 1      # it does sample
    # a huge penalty forbids (positive-logit) repeats of prompt or generated ids: the continuation has no duplicate and differs from greedy if greedy repeats
    pen = run(repetition_penalty=1e6)
    toks = pen.split()
    assert len(set(toks)) == len(toks)
    assert run(num_beams=4).count("<tok_") >= 1                                            # deterministic beam search (tests/test_beam_search.py)
    assert run(num_beams=4, do_sample=True, seed=5) == run(num_beams=4, do_sample=True, seed=5)    # beam sampling, reproducible by seed
    m.generation_defaults = {"do_sample": True, "temperature": 5.0}                      # generation_config.json of a checkpoint
    assert len({run(seed=s) for s in range(4)}) > 1 and run(do_sample=False) == greedy   # explicit kwargs win


def test_consumer_failure_does_not_leave_a_producer_thread_behind():
    """If the group loop raises (here: a KV arena too small for the second group), the producer thread — possibly parked on the ring's
    semaphore or on a full queue — must be told and end, not wait for a slot nobody will release (a leaked thread per failed request
    in a long-lived process)."""
    import threading
    import time
    import lvu
    from quickvideo_amd.engine import QuickPrefillEngine
    from quickvideo_amd.lvu import load_native_model
    from quickvideo_amd.pipeline import PrefillPipeline, _Producer
    from quickvideo_amd.processor import SyntheticProcessor
    m = load_native_model("synthetic:tiny", device="cpu")
    pipe = PrefillPipeline(m, lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=2, num_frames=24), SyntheticProcessor(m.spec),
                           ops=OracleOps())
    calls, orig = [0], QuickPrefillEngine.prefill_group

    def failing(self, *a, **kw):
        calls[0] += 1
        if calls[0] == 2:
            raise RuntimeError("KV arena overflow (injected)")
        return orig(self, *a, **kw)
    QuickPrefillEngine.prefill_group = failing
    before = {t.ident for t in threading.enumerate()}
    try:
        with pytest.raises(RuntimeError, match="injected"):
            pipe.generate("What?", "synthetic://?frames=96&h=56&w=84&seed=2&pattern=gradient", max_new_tokens=1)
    finally:
        QuickPrefillEngine.prefill_group = orig
    deadline = time.time() + 5
    while time.time() < deadline and any(isinstance(t, _Producer) and t.is_alive() for t in threading.enumerate()):
        time.sleep(0.05)
    assert not any(isinstance(t, _Producer) and t.is_alive() for t in threading.enumerate() if t.ident not in before)


def test_image_folder_reader_decodes_and_runs_end_to_end(tmp_path):
    """A video as a directory of JPEG frames (a real decode workload without FFmpeg): the reader decodes + LANCZOS-resizes on worker
    threads, any thread count gives the same pixels, and LVU.generate runs on it (planned from the images' own size)."""
    from PIL import Image
    import lvu
    from quickvideo_amd.frames import ImageFolderVideoReader, open_video
    from quickvideo_amd.lvu import load_native_model
    rs = np.random.RandomState(0)
    base = rs.randint(0, 256, (120, 180, 3), dtype=np.uint8)
    for i in range(24):
        Image.fromarray(np.roll(base, 5 * i, axis=1)).save(tmp_path / f"frame_{i:05d}.jpg", quality=90)
    (tmp_path / "fps.txt").write_text("4.0")
    r = open_video(str(tmp_path), num_threads=1)
    assert isinstance(r, ImageFolderVideoReader) and len(r) == 24 and r.get_fps() == 4.0 and (r.src_h, r.src_w) == (120, 180)
    r.height, r.width, r.frame_iter = 56, 84, 4
    r.process(np.arange(0, 24, 3))
    a = torch.cat([next(r), next(r)])
    r4 = open_video(str(tmp_path), num_threads=4)
    r4.height, r4.width, r4.frame_iter = 56, 84, 8
    r4.process(np.arange(0, 24, 3))
    assert torch.equal(next(r4), a) and a.shape == (8, 3, 56, 84)
    want = np.asarray(Image.open(tmp_path / "frame_00003.jpg").convert("RGB").resize((84, 56), Image.LANCZOS)).transpose(2, 0, 1)
    assert np.array_equal(a[1].numpy(), want)
    m = load_native_model("synthetic:tiny", device="cpu")
    obj = lvu.LVU(lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=4, num_frames=8), model=m)
    obj._ops = OracleOps()
    out = obj.generate("What moves?", str(tmp_path), max_new_tokens=2, eos_token_id=None)
    assert len(out) == 1 and obj._pipeline.last_timings.groups == 2


def test_animated_gif_is_a_video(tmp_path):
    """f3 (frame source): an animated GIF (the container PIL decodes by itself) as `video_path` — frame count, frame rate from the per-frame
    duration, sequential decode of the delta-coded frames, LANCZOS resize to the planned size, end to end through LVU.generate."""
    import lvu
    from PIL import Image
    from quickvideo_amd.frames import open_video
    from quickvideo_amd.lvu import load_native_model
    rs = np.random.RandomState(0)
    frames = [Image.fromarray((rs.rand(60, 90, 3) * 255).astype(np.uint8)).convert("P", palette=Image.ADAPTIVE) for _ in range(24)]
    path = str(tmp_path / "clip.gif")
    frames[0].save(path, save_all=True, append_images=frames[1:], duration=100, loop=0)      # 10 fps
    r = open_video(path)
    assert len(r) == 24 and abs(r.get_fps() - 10.0) < 1e-6 and (r.src_h, r.src_w) == (60, 90)
    r.height, r.width, r.frame_iter = 56, 84, 4
    r.process(np.array([0, 5, 9, 23]))
    got = next(r)
    assert got.shape == (4, 3, 56, 84) and got.dtype == torch.uint8
    with Image.open(path) as im:                                                              # frame 9 decoded independently
        im.seek(9)
        want = np.asarray(im.convert("RGB").resize((84, 56), Image.LANCZOS)).transpose(2, 0, 1)
    assert np.array_equal(got[2].numpy(), want)
    m = load_native_model("synthetic:tiny", device="cpu")
    obj = lvu.LVU(lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=4, num_frames=8), model=m)
    obj._ops = OracleOps()
    out = obj.generate("What is shown?", path, max_new_tokens=2)
    assert out[0].count("<tok_") == 2 and obj._pipeline.last_timings.groups == 2
