"""The multi-GPU layouts BEHIND THE PLUGIN: one process per rank (gloo, CPU tensors, the oracle-backed ops double),
every rank builds `LVU(config, model_init_kwargs={"parallel": mode})` and calls `generate()` like a single-GPU user; rank 0 owns the
frame source, frames are scattered by frame pair, every rank runs the ViT on its share, one all-gather assembles the features, the
engine gets its process groups from the pipeline, the first token comes back from the rank that holds the logits and decode runs on
every layout.  The answer must be the single-process answer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle_ops import OracleOps

QUESTION = "What happens in the video?"
# 112x168 frames -> 8x12 patches -> 24 tokens per frame pair; 48 frames in groups of 12 (24): 144 (288) tokens per group, enough for the
# group-token parallel split to be ACTIVE on 2 (4) ranks (>= 64 rows per rank); ragged variants exercise the padded scatter
VIDEO = "synthetic://?frames=96&h=112&w=168&fps=2&seed=3"


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _cfg(lvu, gs, nframes, **kw):
    return lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=gs, num_frames=nframes, **kw)


def _run(obj, nframes_unused=None, mnt=4):
    out = obj.generate(QUESTION, VIDEO, max_new_tokens=mnt)
    pipe = obj._pipeline
    return out, pipe.last_layout, list(pipe.model.engine.arena.len), pipe.last_timings


def _single(ret, gs, nframes, kw):
    import lvu
    torch.set_num_threads(2)
    obj = lvu.LVU(_cfg(lvu, gs, nframes, **kw), model_init_kwargs={"device": "cpu", "seed": 3})
    obj._ops = OracleOps()
    out, layout, lens, tm = _run(obj)
    ret["single"] = (out, layout, lens, tm.groups, tm.tokens)


def _worker(rank, world, port, mode, gs, nframes, kw, ret):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(2)
        import lvu
        obj = lvu.LVU(_cfg(lvu, gs, nframes, **kw), model_init_kwargs={"device": "cpu", "seed": 3, "parallel": mode})
        obj._ops = OracleOps()
        out, layout, lens, tm = _run(obj)
        eng = obj._pipeline.model.engine
        ret[f"r{rank}"] = (out, layout, lens, tm.groups, tm.tokens, (eng.l0, len(eng.w.layers)), eng.hkv, eng.hq)
        # a second video through the same objects (engine reuse, ring reuse, the cached front-end group)
        out2 = obj.generate(QUESTION, VIDEO, max_new_tokens=2)
        ret[f"again{rank}"] = out2
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:
        import traceback
        ret[f"error{rank}"] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
        raise


def _launch(world, mode, gs=12, nframes=48, **kw):
    ret = mp.Manager().dict()
    _single(ret, gs, nframes, kw)
    mp.spawn(_worker, args=(world, _free_port(), mode, gs, nframes, kw, ret), nprocs=world, join=True)
    for r in range(world):
        assert f"error{r}" not in ret, ret.get(f"error{r}")
    return ret


@pytest.mark.parametrize("world,mode,gs,nframes,layout", [
    (2, "tp", 12, 48, "tp2"), (2, "sp", 12, 48, "pp1xsp2"), (2, "pp", 12, 48, "pp2xsp1"), (3, "pp", 12, 48, "pp3xsp1"),
    (4, "tp", 24, 96, "tp4"), (4, "sp", 24, 96, "pp1xsp4"),
    (2, "sp", 12, 40, "pp1xsp2"),          # short last group: 40 = 12+12+12+4 frames
    (4, "sp", 12, 48, "pp1xsp4"),          # 6 frame pairs on 4 ranks: padded scatter (2, 2, 2, 0 pairs), rows below the sp threshold
    (4, "pp", 12, 48, None),               # more ranks than layers: refused loudly
])
def test_generate_over_ranks_equals_single_process(world, mode, gs, nframes, layout):
    if layout is None:
        ret = mp.Manager().dict()
        with pytest.raises(Exception):
            mp.spawn(_worker, args=(world, _free_port(), mode, gs, nframes, {}, ret), nprocs=world, join=True)
        assert any("needs at least that many layers" in str(ret.get(f"error{r}", "")) for r in range(world))
        return
    ret = _launch(world, mode, gs, nframes)
    out, lay1, lens, groups, tokens = ret["single"]
    assert lay1 == "single" and len(out) == 1 and out[0].count("<tok_") == 4
    for r in range(world):
        o, lay, ln, g, t, (l0, nl), hkv, hq = ret[f"r{r}"]
        assert lay == layout
        assert o == out, (r, o, out)                                           # every rank returns the single-process answer
        assert (g, t) == (groups, tokens)
        if mode == "pp":                                                       # a stage holds ITS layers' cache only
            assert ln == lens[l0:l0 + nl], (r, ln, lens)
        else:
            assert ln == lens
        assert ret[f"again{r}"] == [" ".join(out[0].split(" ")[:2])]
    if mode == "pp":
        assert sorted(ret[f"r{r}"][5][0] for r in range(world)) == sorted({ret[f"r{r}"][5][0] for r in range(world)})   # disjoint stages
        assert sum(ret[f"r{r}"][5][1] for r in range(world)) == 3


def test_auto_layout_picks_a_grid_per_video_and_matches():
    """`parallel="auto"` (opt-in since round 5; the default of a multi-rank job is "tp"): full replica per rank, grid chosen per video from the group count
    (parallel.choose_layout; 4 ranks, 3 layers, 4 groups -> pp2 x sp2 or pp1 x sp4 by the efficiency table)."""
    from quickvideo_amd.parallel import choose_layout, sp_efficiency_table
    want = choose_layout(4, 4, sp_efficiency_table(), 3)
    ret = _launch(4, "auto", 24, 96)
    out = ret["single"][0]
    for r in range(4):
        assert ret[f"r{r}"][1] == f"pp{want[0]}xsp{want[1]}"
        assert ret[f"r{r}"][0] == out


def test_hidden_state_pruning_and_sampling_through_the_pipeline_stages():
    """prefill_prune_starting_layer (hidden rows shrink between stages) + a sampled decode (the deciding rank's choice is broadcast:
    ranks cannot drift apart) through two pipeline stages."""
    ret = _launch(2, "pp", 12, 48, prefill_prune_starting_layer=1)
    out = ret["single"][0]
    assert ret["r0"][0] == out and ret["r1"][0] == out
    # ... and under group-token parallelism (round 4): segments that prune the hidden rows run replicated on every sp rank (the surviving
    # rows are an irregular subset the zigzag K/V exchange cannot follow), the prompt tail as before — same answer, same cache lengths
    ret = _launch(2, "sp", 12, 48, prefill_prune_starting_layer=1)
    assert ret["r0"][0] == ret["r1"][0] == ret["single"][0] and ret["r0"][2] == ret["r1"][2] == ret["single"][2]


@pytest.mark.parametrize("world,mode", [(2, "sp"), (2, "tp"), (2, "pp"), (4, "auto")])
def test_query_score_pruning_through_the_plugin_on_every_layout(world, mode):
    """top_k_predict_type="query_attention_weights" (the prompt rows ride along with every group and score its keys) behind the plugin:
    tensor parallel (per-head sums all-gathered), layer pipeline (n + m rows travel), group-token parallel (these segments run replicated)
    and the auto grid — the single-process answer on every rank."""
    gs, nf = (12, 48) if world == 2 else (24, 96)
    ret = _launch(world, mode, gs, nf, top_k_predict_type="query_attention_weights")
    out = ret["single"][0]
    for r in range(world):
        assert ret[f"r{r}"][0] == out, (r, ret[f"r{r}"][0], out)


def _beam_worker(rank, world, port, mode, ret):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(2)
        import lvu
        obj = lvu.LVU(_cfg(lvu, 12, 48), model_init_kwargs={"device": "cpu", "seed": 3, "parallel": mode})
        obj._ops = OracleOps()
        ret[f"r{rank}"] = obj.generate(QUESTION, VIDEO, max_new_tokens=4, num_beams=3, eos_token_id=-1)
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:
        import traceback
        ret[f"error{rank}"] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
        raise


@pytest.mark.parametrize("mode", ["pp", "sp", "tp"])
def test_beam_search_over_ranks_equals_single_process(mode):
    """Beam search on the multi-GPU layouts: the rank that holds the logits runs the search and tells the others, step by step, which beams to
    extend by which tokens; every rank advances its share of the model.  Same answer as the single-process beam search on every rank."""
    import lvu
    ret = mp.Manager().dict()
    obj = lvu.LVU(_cfg(lvu, 12, 48), model_init_kwargs={"device": "cpu", "seed": 3})
    obj._ops = OracleOps()
    want = obj.generate(QUESTION, VIDEO, max_new_tokens=4, num_beams=3, eos_token_id=-1)
    assert want[0].count("<tok_") == 4
    mp.spawn(_beam_worker, args=(2, _free_port(), mode, ret), nprocs=2, join=True)
    for r in range(2):
        assert f"error{r}" not in ret, ret.get(f"error{r}")
        assert ret[f"r{r}"] == want, (mode, r, ret[f"r{r}"], want)


def test_forced_grid_env(monkeypatch):
    from quickvideo_amd.parallel import ParallelContext
    ctx = ParallelContext("auto", 8, 3)
    monkeypatch.setenv("QP_GRID", "4x2")
    assert ctx.grid(4, 28) == (4, 2) and ParallelContext("sp", 8, 0).grid(4, 28) == (1, 8)
    monkeypatch.setenv("QP_GRID", "4x4")
    with pytest.raises(ValueError):
        ctx.grid(4, 28)


def test_run_ahead_gate_model():
    """tools/sim_pipeline.py (dependency model of the layer pipeline): with every stage gating the ViT run-ahead on its own prefill a deep
    pipe is throttled towards 2/(pp-1); with rank 0's gate alone (pipeline.py) it runs at full rate."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sim_pipeline", os.path.join(os.path.dirname(__file__), "..", "tools", "sim_pipeline.py"))
    sim = importlib.util.module_from_spec(spec); spec.loader.exec_module(sim)
    for pp in (2, 3, 4, 8):
        assert sim.rate(pp) == pytest.approx(1.0, abs=0.01)
    assert sim.rate(4, every_stage_gates=True) < 0.7 and sim.rate(8, every_stage_gates=True) < 0.3
    assert sim.rate(2, every_stage_gates=True) == pytest.approx(1.0, abs=0.01)


def test_multi_gpu_runtime_defaults_do_not_override_the_user(monkeypatch):
    from quickvideo_amd.parallel import multi_gpu_runtime_defaults
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.delenv("QP_IPC_DMABUF", raising=False)
    multi_gpu_runtime_defaults()
    # the dmabuf-IPC switch belongs to the host driver: exported only on request, never for every user of the package
    assert os.environ["GPU_MAX_HW_QUEUES"] == "8" and "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ
    multi_gpu_runtime_defaults(ipc_dmabuf=True)
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.setenv("QP_IPC_DMABUF", "1")
    multi_gpu_runtime_defaults()
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "2")
    multi_gpu_runtime_defaults(ipc_dmabuf=True)
    assert os.environ["GPU_MAX_HW_QUEUES"] == "2" and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"


def test_multi_rank_default_layout_is_the_contract_one_and_groups_carry_a_timeout(monkeypatch):
    """A multi-rank job that does not say otherwise gets tensor parallelism (collectives on the job's own group only); sp / pp / auto are
    opt-in until a multi-GPU box has run them, and every process group the package creates carries a timeout (a peer that never arrives is
    an error after QP_DIST_TIMEOUT_S, not a hang)."""
    import inspect
    from quickvideo_amd import parallel
    src = inspect.getsource(parallel)
    assert src.count("new_group(") == src.count("timeout=dist_timeout()") - 1 >= 2      # every new_group + init_process_group
    monkeypatch.setenv("QP_DIST_TIMEOUT_S", "12.5")
    assert parallel.dist_timeout().total_seconds() == 12.5
    monkeypatch.delenv("QP_PARALLEL", raising=False)

    class FakeDist:
        @staticmethod
        def is_available(): return True
        @staticmethod
        def is_initialized(): return True
        @staticmethod
        def get_world_size(group=None): return 4
        @staticmethod
        def get_rank(group=None): return 1
    monkeypatch.setattr(parallel.torch, "distributed", FakeDist)
    assert parallel.resolve().mode == "tp" and parallel.resolve("auto").mode == "auto"
    monkeypatch.setenv("QP_PARALLEL", "pp")
    assert parallel.resolve().mode == "pp"


def test_layout_cost_model():
    from quickvideo_amd.parallel import choose_layout, layout_efficiency, stage_balance
    assert stage_balance(28, 8) == pytest.approx(3.5 / 4) and stage_balance(28, 4) == 1.0 and stage_balance(80, 8) == 1.0
    eff = {1: 1.0, 2: 0.93, 4: 0.82, 8: 0.62}
    # the 1-hour video (450 groups) on 8 GPUs: an 8-stage pipe of 28 layers is capped at 7/8 by its 4-layer stages; 4 stages x 2-rank
    # row groups (7 layers each, balanced) wins
    assert choose_layout(450, 8, eff, 28) == (4, 2)
    assert layout_efficiency(450, 8, 1, eff, 28) == pytest.approx(450 / 457 * 0.875)
    assert choose_layout(450, 8, eff, 80) == (8, 1)                        # 72B: 10 layers per stage
    assert choose_layout(4, 8, eff, 28)[0] <= 2                            # 4 groups: no deep pipe
    assert choose_layout(450, 4, eff, 28) == (4, 1) and choose_layout(450, 2, eff, 28) == (2, 1)
