"""What a ONE-GPU box can check of the multi-GPU path (the reference itself has no collectives, lvu/lvu.py:13):

* every collective the engine issues, through torch.distributed's backend "nccl" (= RCCL on ROCm) with world_size 1 — dtype, shape,
  stream and device-binding errors surface without a second GPU;
* tensor parallelism over two PROCESSES sharing the GPU (gloo moves the bytes, the kernels are the real ones) at the
  Qwen2-VL-72B / TP=8 per-rank dims: the ranks derive the kept-index lists of the single-process engine.

Each test runs its ranks in spawned processes (the pytest process never owns a process group)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny_case():
    from oracle import qp_oracle as O
    from quickvideo_amd import planner
    so = O.TextSpec(hidden=512, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=1024, n_layers=3, vocab=320)
    w = {k: v.to(torch.bfloat16) for k, v in O.synthetic_text_weights(so, seed=7, norm_jitter=0.1).items()}
    frames, gh, gw, gs, prefix, tail = 24, 16, 24, 8, 15, 20               # 3 groups x 384 tokens (>= 64: the sp path is taken)
    T = prefix + (frames // 2) * (gh // 2) * (gw // 2) + tail
    plan = planner.plan_groups(frames, gs, gh, gw, prefix, T)
    pos, _ = planner.mrope_positions(prefix, (frames // 2, gh, gw), tail)
    embeds = torch.from_numpy(np.random.RandomState(1).standard_normal((T, 512)).astype(np.float32) * 0.5).to(torch.bfloat16)
    return so, w, plan, pos, embeds, T


def _run_engine(eng, plan, pos, embeds):
    e, p = embeds.cuda(), torch.from_numpy(pos).cuda()
    eng.kept_trace = []
    start = 0
    for n in plan.tokens:
        eng.prefill_group(e[start:start + n], p[:, start:start + n]); start += n
    logits = eng.prefill_tail(e[start:], p[:, start:])
    torch.cuda.synchronize()
    kept = [None if k is None else k.cpu().numpy().copy() for _, k in eng.kept_trace]
    return logits.float().cpu().numpy(), kept, list(eng.arena.len)


def _nccl_world1_worker(rank, port, ret):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        from quickvideo_amd.engine import QuickPrefillEngine
        from quickvideo_amd.lvu_config import LVUConfig
        from quickvideo_amd.spec import TextSpec
        from quickvideo_amd.weights import DecoderWeights
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)            # bench.py's call (nccl == RCCL on ROCm)
        ret["backend"] = dist.get_backend()
        so, w, plan, pos, embeds, T = _tiny_case()
        spec = TextSpec(hidden=512, n_heads=4, n_kv_heads=2, head_dim=128, intermediate=1024, n_layers=3, vocab=320)
        cfg = LVUConfig("x", top_p=0.5, video_group_size=8)
        mk = lambda **kw: QuickPrefillEngine(DecoderWeights.from_named(spec, w, dev), cfg, capacity=T + 8,          # noqa: E731
                                             max_group_tokens=max(plan.tokens + [plan.tail_len]), device=dev, **kw)
        os.environ["QP_NATIVE_SEGMENT"] = "0"       # the bit-for-bit comparison below is between two runs of the per-operator loop (same GEMM calls)
        # ... and with the GEMM decompositions NOT timed: the plain engine picks row splits / hipBLASLt candidates by stopwatch (a > 3 % win),
        # the TP projections (_linear_reduced) take fixed row blocks — on these tiny shapes timing noise decides whether the two run the same
        # kernels, and the logits then agreed bit for bit in 5 runs of 7 (another algorithm = another fp32 accumulation order)
        os.environ["QP_TUNE_GEMMS"] = "0"
        base = _run_engine(mk(), plan, pos, embeds)
        # tensor parallel layout on a 1-rank RCCL group: 2 x all_reduce bf16 [n, d] + all_gather_into_tensor fp32 [Hkv, n] per layer
        grp = dist.new_group(ranks=[0])                                                  # bench.py builds stage groups like this
        os.environ["QP_TP_CHUNKS"] = "1"                                                   # unsplit: bit-comparable with the plain engine
        tp = _run_engine(mk(tp_group=grp), plan, pos, embeds)
        # round 4: the projections in two row blocks, block 0's all-reduce ASYNCHRONOUS on RCCL's stream beside block 1's GEMM
        os.environ["QP_TP_CHUNKS"] = "2"
        calls, real = [], dist.all_reduce
        dist.all_reduce = lambda t_, *a, **kw: (calls.append((tuple(t_.shape), bool(kw.get("async_op", False)))), real(t_, *a, **kw))[1]
        try:
            tp2 = _run_engine(mk(tp_group=grp), plan, pos, embeds)
        finally:
            dist.all_reduce = real
            os.environ.pop("QP_TP_CHUNKS")
        ret["tp2"], ret["tp2_calls"] = tp2, calls
        # group-token parallel layout: all_gather_into_tensor of the uint8 K|V|sums exchange block + qp_sp_unpack per layer
        sp = _run_engine(mk(sp_group=grp, sp_rank=0, sp_size=1), plan, pos, embeds)
        # the remaining calls bench.py makes around the engine
        tok = torch.tensor([int(np.argmax(base[0]))], device=dev)
        dist.broadcast(tok, src=0)
        t = torch.tensor([1.5], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        ret["base"], ret["tp"], ret["sp"], ret["tok"], ret["max"] = base, tp, sp, int(tok.item()), float(t.item())
        dist.destroy_process_group()
    except BaseException as e:                                                          # surfaces in the parent's assertion
        import traceback
        ret["error"] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
        raise


def test_every_engine_collective_on_nccl_world_size_1():
    """RCCL pre-flight (round-2 review): init_process_group("nccl"), new_group, the engine's all_reduce (bf16 [n, d]),
    all_gather_into_tensor (fp32 key sums; the uint8 exchange block of the group-token parallel mode), broadcast, MAX all_reduce and
    barrier all execute on the GPU through RCCL.  A 1-rank collective is the identity, so: the TP-layout engine must reproduce the
    plain engine's kept lists and cache lengths EXACTLY and its logits bit for bit up to the prune path it takes (qp_norm_keys
    instead of the fused RoPE keys: same bits); the sp-layout engine (zigzag rows, two attention launches per layer) must give the
    same kept lists and logits within the attention tolerance."""
    ret = mp.Manager().dict()
    mp.spawn(_nccl_world1_worker, args=(_free_port(), ret), nprocs=1, join=True)
    assert "error" not in ret, ret.get("error")
    assert ret["backend"] == "nccl"
    (l0, k0, n0), (l1, k1, n1), (l2, k2, n2) = ret["base"], ret["tp"], ret["sp"]
    assert n0 == n1 == n2
    for a, b, c in zip(k0, k1, k2):
        assert (a is None) == (b is None) == (c is None)
        if a is not None:
            assert np.array_equal(a, b), "TP layout on a 1-rank group must keep the same tokens"
            assert len(set(a.tolist()) & set(c.tolist())) / len(a) >= 0.98                 # sp: other attention tiling -> rounding
    assert np.array_equal(l0, l1), float(np.max(np.abs(l0 - l1)))
    assert np.max(np.abs(l0 - l2)) <= 4e-2
    # the row-block form with asynchronous all-reduces (the default under tensor parallelism): same tokens kept, logits to GEMM-tiling noise
    l3, k3, n3 = ret["tp2"]
    assert n3 == n0 and np.max(np.abs(l0 - l3)) <= 4e-2
    for a, b in zip(k0, k3):
        assert (a is None) == (b is None) and (a is None or len(set(a.tolist()) & set(b.tolist())) / len(a) >= 0.98)
    blocks = [c for c in ret["tp2_calls"] if c[0][0] >= 128]
    assert blocks and all(is_async for _, is_async in blocks) and len(blocks) == 3 * 3 * 2 * 2     # 3 groups x 3 layers x (o, down) x 2 blocks
    assert ret["tok"] == int(np.argmax(l0)) and ret["max"] == 1.5


# ---------------------------------------------------------------- TP=2 on one GPU (gloo) at the 72B / TP=8 per-rank dims
_SLICE = dict(hidden=8192, head_dim=128, n_layers=2, vocab=512)


def _slice_case(tp):
    """Full model = `tp` ranks of the cfg5 per-rank slice (8 q heads + 1 kv head, 3696 MLP columns each); one group of n = 960."""
    from oracle import qp_oracle as O
    dims = dict(_SLICE, n_heads=8 * tp, n_kv_heads=tp, intermediate=3696 * tp)
    so = O.TextSpec(**dims)
    w = O.hashed_text_weights(so, seed=41, norm_jitter=0.05, device="cuda")        # the same bits in every process
    n = 960
    embeds = O.hashed_normal((n + 24, so.hidden), 42, 0.5, device="cuda")
    pos = np.tile(np.arange(n + 24, dtype=np.int64), (3, 1))
    return dims, w, embeds, pos, n


def _tp_gloo_worker(rank, world, port, ret):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        from quickvideo_amd.engine import QuickPrefillEngine
        from quickvideo_amd.lvu_config import LVUConfig
        from quickvideo_amd.spec import TextSpec
        from quickvideo_amd.weights import DecoderWeights
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dims, w, embeds, pos, n = _slice_case(world)
        spec = TextSpec(**dims)
        cfg = LVUConfig("x", top_p=0.5, video_group_size=16)
        dw = DecoderWeights.from_named(spec, w, "cuda:0", tp_rank=rank, tp_size=world)
        eng = QuickPrefillEngine(dw, cfg, capacity=n + 64, max_group_tokens=n, device="cuda:0", tp_group=dist.group.WORLD)
        eng.kept_trace = []
        e, p = embeds.cuda(), torch.from_numpy(pos).cuda()
        eng.prefill_group(e[:n], p[:, :n])
        torch.cuda.synchronize()
        ret[f"kept{rank}"] = [k.cpu().numpy().copy() for _, k in eng.kept_trace]
        ret[f"heads{rank}"] = (eng.hq, eng.hkv, eng.li)
        # the rank's staged layer-0 keys are not kept; its arena rows of layer 0 are the kept rows of ITS kv head
        ret[f"k0_{rank}"] = eng.arena.k(0)[:, :eng.arena.len[0]].cpu().view(torch.int16).numpy().copy()
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:
        import traceback
        ret[f"error{rank}"] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
        raise


def _single_worker(rank, world, ret):
    try:
        from quickvideo_amd.engine import QuickPrefillEngine
        from quickvideo_amd.lvu_config import LVUConfig
        from quickvideo_amd.spec import TextSpec
        from quickvideo_amd.weights import DecoderWeights
        torch.cuda.set_device(0)
        dims, w, embeds, pos, n = _slice_case(world)
        spec = TextSpec(**dims)
        dw = DecoderWeights.from_named(spec, w, "cuda:0")
        eng = QuickPrefillEngine(dw, LVUConfig("x", top_p=0.5, video_group_size=16), capacity=n + 64, max_group_tokens=n, device="cuda:0")
        eng.kept_trace = []
        e, p = embeds.cuda(), torch.from_numpy(pos).cuda()
        eng.prefill_group(e[:n], p[:, :n])
        torch.cuda.synchronize()
        ret["kept"] = [k.cpu().numpy().copy() for _, k in eng.kept_trace]
        ret["k0"] = eng.arena.k(0)[:, :eng.arena.len[0]].cpu().view(torch.int16).numpy().copy()
    except BaseException as e:
        import traceback
        ret["error"] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
        raise


def test_tp2_ranks_on_one_gpu_reproduce_single_process_kept_lists():
    """"Key norms are bit-stable across TP degrees"  on the HIP path: two ranks, each holding the cfg5 per-rank slice
    (d = 8192, 8 q + 1 kv head, I/8 = 3696 columns), share the GPU and exchange through gloo; n = 960, rho = 0.5.
    Layer 0 sees the same input rows on every layout, its key rows come from the same weights, and the per-head sums are
    all-gathered and added in head order — so BOTH ranks must derive the single-process engine's layer-0 kept list exactly
    (provided the key rows themselves are bit-identical, which is asserted first: a rank's kept rows == the single-process arena
    rows of that kv head).  Layer 1 sits behind a bf16 all-reduce (different summation order than one GEMM): overlap bar only."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_single_worker, args=(world, ret), nprocs=1, join=True)
    assert "error" not in ret, ret.get("error")
    mp.spawn(_tp_gloo_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        assert f"error{r}" not in ret, ret.get(f"error{r}")
        assert ret[f"heads{r}"] == (8, 1, 3712)                  # 3696 columns stored zero-padded to 29 x 128 (weights.padded_inter, round 6)
    ref, k0 = ret["kept"], ret["k0"]
    for a, b in zip(ret["kept0"], ret["kept1"]):
        assert np.array_equal(a, b), "ranks disagree on the kept tokens"                      # identical on every rank, every layer
    same_rows = all(np.array_equal(ret[f"k0_{r}"][0], k0[r]) for r in range(world))
    ov0 = len(set(ret["kept0"][0].tolist()) & set(ref[0].tolist())) / len(ref[0])
    ov1 = len(set(ret["kept0"][1].tolist()) & set(ref[1].tolist())) / len(ref[1])
    print(f"layer-0 kept rows bit-identical to the single-process arena: {same_rows}; kept-list overlap layer 0 {ov0:.4f}, layer 1 {ov1:.4f}")
    assert np.array_equal(ret["kept0"][0], ref[0]), f"layer-0 kept list differs from the single-process engine (overlap {ov0:.4f})"
    assert same_rows
    assert ov1 >= 0.95, ov1


# ---------------------------------------------------------------- round 4: the layouts behind the plugin, two ranks sharing the GPU
_PLUGIN_VIDEO = "synthetic://?frames=96&h=112&w=168&fps=2&seed=3"


def _plugin_worker(rank, world, port, mode, ret):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        torch.cuda.set_device(0)
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        import lvu
        cfg = lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=12, num_frames=48)
        obj = lvu.LVU(cfg, model_init_kwargs={"device": "cuda:0", "seed": 3, "parallel": mode})
        out = obj.generate("What happens in the video?", _PLUGIN_VIDEO, max_new_tokens=4)
        pipe = obj._pipeline
        torch.cuda.synchronize()
        ret[f"{mode}{rank}"] = (out, pipe.last_layout, list(pipe.model.engine.arena.len), pipe.last_timings.ttft)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    except BaseException as e:
        import traceback
        ret[f"error_{mode}{rank}"] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
        raise


def test_plugin_generate_two_ranks_on_one_gpu_equals_single_process():
    """`LVU(..., model_init_kwargs={"parallel": mode}).generate()` on TWO ranks sharing the GPU (gloo) reproduces
    the single-process answer THROUGH THE PIPELINE — rank-0 producer + pinned ring, frames to the other rank, data-parallel ViT on the
    HIP kernels + all-gather, the layout's engine with its groups, the deciding rank's token broadcast, eager decode on every layout.
    pp / sp move rows between identical replicas (same kernels, same numbers up to the attention tiling): tokens equal.  tp sums
    head / column partials in a different order (bf16 all-reduce): the FIRST token must match, the answer is compared and reported."""
    ret = mp.Manager().dict()
    mp.spawn(_plugin_worker, args=(1, _free_port(), "single", ret), nprocs=1, join=True)
    assert "error_single0" not in ret, ret.get("error_single0")
    out1, lay1, lens1, _ = ret["single0"]
    assert lay1 == "single" and out1[0].count("<tok_") == 4
    for mode, layout in (("pp", "pp2xsp1"), ("sp", "pp1xsp2"), ("tp", "tp2")):
        mp.spawn(_plugin_worker, args=(2, _free_port(), mode, ret), nprocs=2, join=True)
        for r in range(2):
            assert f"error_{mode}{r}" not in ret, ret.get(f"error_{mode}{r}")
            out, lay, lens, ttft = ret[f"{mode}{r}"]
            assert lay == layout and ttft > 0
            assert out == ret[f"{mode}0"][0]                                    # both ranks return the same answer
            assert out[0].split(" ")[0] == out1[0].split(" ")[0], (mode, out, out1)   # the first token is the single-process one
            if mode != "tp":
                assert out == out1, (mode, out, out1)
            if mode == "sp":
                assert lens == lens1
        print(mode, ret[f"{mode}0"][0], "single:", out1)
