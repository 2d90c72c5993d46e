"""Multi-GPU layouts of the QuickPrefill path, selected behind the plugin (north_star: "video-in -> first-token latency
... at 1, 2, 4 and 8 MI355X").

The reference spans the visible GPUs with `device_map="auto"` (lvu/lvu.py:11-16: HF accelerate places consecutive decoder layers on
consecutive devices and the patched layer hops devices at lvu/models/qwen25_lvu.py:182) — ONE process, layers executed one device after
the other.  Here: one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI), every rank constructs the same
`LVU(config, model_init_kwargs={"parallel": ...})` and calls the same `generate()`; rank 0 owns the frame source.

    tp    tensor parallel over heads / MLP columns (the north_star contract): weights and KV sharded at load; per layer two [n, d]
          all-reduces + one all-gather of the fp32 key sums (engine.py).
    sp    group-token parallel: weights + KV replicated, every rank takes a zigzag slice of each group's rows; one K/V all-gather per layer.
    pp    layer pipeline: rank r holds a contiguous slice of the layers and their KV; one [n, d] hand-off per group and stage.
    auto  a pp x sp grid chosen per VIDEO from the group count (choose_layout): long videos pipeline, short ones split rows.

For sp / pp / auto every rank holds the full weight set (15 GB of the 288 GB for the 7B model), so the grid is a per-video decision and
a pipeline stage is a VIEW of the replica (stage_weights).  In every layout the vision tower runs data-parallel over the frame pairs
of a group — each temporal patch is its own attention sequence, so the split is exact — followed by one all-gather of the [n, d]
features (pipeline.py::PrefillPipeline._vit_parallel).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch

from .weights import DecoderWeights, pp_layer_split

MODES = ("single", "tp", "sp", "pp", "auto")


def dist_timeout():
    """Timeout of every process group this package creates (QP_DIST_TIMEOUT_S, default 600 s): a collective, a pipeline hand-off or a
    front-end scatter whose peer never arrives — e.g. communicators issued in a rank-dependent order that the runtime cannot keep
    co-resident — then ABORTS the job with torch's NCCL-watchdog error naming the collective, instead of hanging it.  (No multi-GPU box
    has executed the sp / pp / auto layouts yet: a mis-ordered communicator must show up as an error.)"""
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get("QP_DIST_TIMEOUT_S", "600")))

# Prior for "auto" when nobody measured this machine: efficiency of an s-rank group-token parallel group relative to one GPU (GEMMs at
# M = n / s, the K/V all-gather, the replicated prune).  bench.py measures the table at start-up (probe_sp_efficiency) and hands it to
# the pipeline; QP_SP_EFFICIENCY="2:0.93,4:0.8,8:0.6" overrides it.
DEFAULT_SP_EFFICIENCY = {1: 1.0, 2: 0.93, 4: 0.82, 8: 0.62}


def sp_efficiency_table(override: Optional[Dict[int, float]] = None) -> Dict[int, float]:
    if override:
        return {int(k): float(v) for k, v in override.items()}
    env = os.environ.get("QP_SP_EFFICIENCY")
    if env:
        t = {1: 1.0}
        t.update({int(a): float(b) for a, b in (item.split(":") for item in env.split(",") if item)})
        return t
    return dict(DEFAULT_SP_EFFICIENCY)


def stage_balance(n_layers: int, pp: int) -> float:
    """Average / maximum layers per pipeline stage: the pipe runs at the pace of its heaviest stage (28 layers over 8 stages: 3.5 / 4)."""
    return (n_layers / pp) / max(l1 - l0 for l0, l1 in (pp_layer_split(n_layers, pp, s) for s in range(pp)))


def choose_layout(n_groups: int, world: int, eff_sp: Dict[int, float], n_layers: Optional[int] = None) -> Tuple[int, int]:
    """(pp, sp) with pp * sp == world: a layer pipeline of pp stages, each a group-token parallel group of sp ranks.  Cost model: the
    pipe is busy G / (G + pp - 1) of the time (fill + drain), runs at the pace of its heaviest stage (stage_balance; round 4 — an
    8-stage split of 28 layers is capped at 7/8 before anything else), and an sp group of s ranks runs at eff_sp[s] of one GPU.  The ViT,
    the embedding gather and the prompt tail's lm_head are NOT in the stage model on purpose: the tower is data-parallel over all ranks
    in every layout (equal share per rank), the other two are a few rows once per video."""
    best, best_eff = (world, 1), -1.0
    pp = 1
    while pp <= world:
        sp = world // pp
        if pp * sp == world and sp in eff_sp and (n_layers is None or pp <= n_layers):
            eff = n_groups / (n_groups + pp - 1) * eff_sp[sp] * (stage_balance(n_layers, pp) if n_layers else 1.0)
            if eff > best_eff + 1e-9:
                best, best_eff = (pp, sp), eff
        pp *= 2
    return best


def layout_efficiency(n_groups: int, pp: int, sp: int, eff_sp: Dict[int, float], n_layers: int) -> float:
    return n_groups / (n_groups + pp - 1) * eff_sp.get(sp, 0.0) * stage_balance(n_layers, pp)


def stage_weights(w: DecoderWeights, pp: int, stage: int) -> DecoderWeights:
    """Pipeline stage `stage` of `pp` as a view of a full replica (no copy): its contiguous layer slice + embedding / norm / lm_head."""
    if pp == 1:
        return w
    assert w.layer0 == 0 and len(w.layers) == w.n_layers_total, "stage views are cut from a full replica"
    l0, l1 = pp_layer_split(w.n_layers_total, pp, stage)
    return DecoderWeights(w.spec, w.embed, w.layers[l0:l1], w.norm, w.lm_head, w.tp_rank, w.tp_size, l0, w.n_layers_total)


@dataclass
class ParallelContext:
    """What `LVU(..., model_init_kwargs={"parallel": mode})` resolves to: the request, this process's place in the job and the
    process group the job's collectives run on.  `mode == "single"`: no torch.distributed call is ever made."""
    mode: str = "single"
    world: int = 1
    rank: int = 0
    group: object = None                     # None = the default group
    sp_efficiency: Optional[Dict[int, float]] = None
    grid_override: Optional[Tuple[int, int]] = None   # (pp, sp) forced for the next videos (bench.py: warm-up clip on the main video's grid)
    _subgroups: dict = field(default_factory=dict)

    @property
    def on(self) -> bool:
        return self.mode != "single" and self.world > 1

    def grid(self, n_groups: int, n_layers: int) -> Tuple[int, int]:
        """(pp, sp) for a video of n_groups groups under this context's mode (tp / single: (1, 1))."""
        if not self.on or self.mode == "tp":
            return 1, 1
        if self.grid_override is not None:
            return self.grid_override
        if self.mode == "sp":
            return 1, self.world
        forced = os.environ.get("QP_GRID")                              # "4x2": pp x sp forced for mode "auto" (A/B of the cost model's choice)
        if forced and self.mode == "auto":
            pp, sp = (int(v) for v in forced.lower().split("x"))
            if pp * sp != self.world or pp > n_layers:
                raise ValueError(f"QP_GRID={forced!r}: pp * sp must equal the {self.world} ranks of the job and pp <= {n_layers} layers")
            return pp, sp
        if self.mode == "pp":
            if self.world > n_layers:
                raise ValueError(f"layer pipeline over {self.world} ranks needs at least that many layers (model has {n_layers})")
            return self.world, 1
        return choose_layout(n_groups, self.world, sp_efficiency_table(self.sp_efficiency), n_layers)

    def global_rank(self, r: int) -> int:
        return torch.distributed.get_global_rank(self.group, r) if self.group is not None else r

    def stage_group(self, pp: int, sp: int):
        """This rank's sp group inside a pp x sp grid (ranks of a stage are consecutive).  new_group is collective over the job, so
        EVERY rank creates EVERY stage's group, in the same order, once per grid shape."""
        if sp == 1:
            return None
        if sp == self.world:
            return self.group if self.group is not None else torch.distributed.group.WORLD
        key = (pp, sp)
        if key not in self._subgroups:
            gs = [torch.distributed.new_group(ranks=[self.global_rank(s * sp + i) for i in range(sp)], timeout=dist_timeout()) for s in range(pp)]
            self._subgroups[key] = gs
        return self._subgroups[key][self.rank // sp]

    def pair_groups(self, pp: int, sp: int):
        """(group towards the next stage, group from the previous stage) of this rank in a pp x sp grid: one 2-rank process group per
        adjacent pair of pipeline counterparts = one RCCL communicator and stream per DIRECTION of a stage.  On a communicator that was
        created eagerly (init_process_group(device_id=...)) torch treats unbatched send/recv as collectives of that communicator and
        serialises them with everything else on it (its own warning, seen on the GPU box: profiles/r4_rccl_self_p2p_probe.txt) — a stage's
        recv of segment g+1 would queue behind its still unmatched send of segment g.  Collective creation: every rank creates every
        pair's group in the same order, once per grid shape."""
        if pp == 1:
            return None, None
        key = ("pairs", pp, sp)
        if key not in self._subgroups:
            self._subgroups[key] = {(s, i): torch.distributed.new_group(ranks=[self.global_rank(s * sp + i), self.global_rank((s + 1) * sp + i)], timeout=dist_timeout())
                                    for s in range(pp - 1) for i in range(sp)}
        g, (stage, i) = self._subgroups[key], (self.rank // sp, self.rank % sp)
        return g.get((stage, i)), g.get((stage - 1, i))

    def engine_kwargs(self, pp: int, sp: int) -> dict:
        """Constructor arguments of QuickPrefillEngine for this rank in a pp x sp grid (the wiring bench.py::measure used to do by hand)."""
        if not self.on:
            return {}
        if self.mode == "tp":
            return {"tp_group": self.group if self.group is not None else torch.distributed.group.WORLD}
        stage, sp_rank = self.rank // sp, self.rank % sp
        kw = {}
        if sp > 1:
            kw.update(sp_group=self.stage_group(pp, sp), sp_rank=sp_rank, sp_size=sp)
        if pp > 1:
            nxt, prv = self.pair_groups(pp, sp)
            kw.update(pp_rank=stage, pp_size=pp, pp_peers=[self.global_rank(s * sp + sp_rank) for s in range(pp)], pp_send_group=nxt, pp_recv_group=prv)
        return kw

    def describe(self, pp: int = 1, sp: int = 1) -> str:
        if not self.on:
            return "single"
        return f"tp{self.world}" if self.mode == "tp" else f"pp{pp}xsp{sp}"


def resolve(parallel: Optional[str] = None, group=None) -> ParallelContext:
    """`parallel`: one of MODES, or None -> $QP_PARALLEL, else "tp" when torch.distributed is initialised with more than one rank and
    "single" otherwise.  A mode other than "single" in a 1-rank job resolves to "single" (the same script runs on 1 and on 8 GPUs).
    The multi-rank default is the contract layout (tensor parallel: all_reduce / all_gather on the job's own group, no
    sub-communicators); "auto" / "sp" / "pp" are OPT-IN (argument or QP_PARALLEL) until a real 2/4/8-GPU run of them is on record —
    they have executed on gloo and with several ranks on one GPU only."""
    mode = parallel or os.environ.get("QP_PARALLEL")
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    if mode is None:
        mode = "tp" if dist_on and torch.distributed.get_world_size(group) > 1 else "single"
    if mode not in MODES:
        raise ValueError(f"parallel={mode!r}: expected one of {MODES}")
    if mode == "single" or not dist_on:
        if mode != "single" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise RuntimeError(f"parallel={mode!r} needs torch.distributed.init_process_group() before the model is built (one process per GPU)")
        return ParallelContext("single")
    world, rank = torch.distributed.get_world_size(group), torch.distributed.get_rank(group)
    if world == 1:
        return ParallelContext("single")
    return ParallelContext(mode, world, rank, group)


def multi_gpu_runtime_defaults(ipc_dmabuf: Optional[bool] = None) -> None:
    """Environment a multi-GPU rank wants BEFORE its first HIP call (setdefault: the user's own setting wins).

    GPU_MAX_HW_QUEUES=8: the HIP runtime multiplexes a process's streams onto 4 hardware queues per priority by default, and streams on
    one queue run in submission order.  A rank of the pipeline owns more than that — the LLM stream, the ViT stream, the copy stream
    (rank 0) and one RCCL stream per process group it talks on (front end, pair-send, pair-recv, stage group, the job's group) — and an
    RCCL kernel that spins for its peer would hold back whatever shares its queue: a hand-off waiting for the NEXT stage would stall THIS
    stage's compute (lock-step instead of a pipeline).  Measured on one MI355X (tools/probe/probe_hw_queues.py,
    profiles/r4_hw_queue_classes.txt): main + 8 default + 6 high-priority streams fall into 8 independent classes at the default, 14 at
    8 (every high-priority stream — RCCL's and the copy stream — and the first 7 default ones on queues of their own) and 15 at 16; the
    single-GPU pipeline runs at the same speed under 16 (cfg4s: 48.47 k vs 48.44-48.59 k tok/s).  8 is enough for a rank's streams and
    keeps the process's queue count (8 per priority) inside what the command processor schedules without time-slicing.
    Deadlock is excluded either way by construction (a stage posts recv(g) only after its part of all-gather(g) is done), this is
    about not serialising.  No multi-GPU box has run it."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC) is a property of the HOST DRIVER, not of this package: exported only on request —
    # `ipc_dmabuf=True` (bench.py, the test launchers: the boxes this repo is measured on support nothing else) or QP_IPC_DMABUF=1
    if ipc_dmabuf is None:
        ipc_dmabuf = os.environ.get("QP_IPC_DMABUF") == "1"
    if ipc_dmabuf:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def init_distributed(backend: Optional[str] = None) -> ParallelContext:
    """Convenience for a script launched with one process per GPU (`python -m torch.distributed.run --nproc-per-node N --master-addr
    127.0.0.1 script.py`): binds this process to its GPU (LOCAL_RANK), initialises torch.distributed (backend "nccl" = RCCL over xGMI on a
    GPU box; "gloo" on CPU) unless that has been done already, and returns the resolved context.  In a plain single-process run it does
    nothing and returns the single context — the same script works on 1 and on 8 GPUs."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not torch.distributed.is_initialized():
        multi_gpu_runtime_defaults()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        cuda = torch.cuda.is_available()
        kw = {}
        if cuda:
            dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
            torch.cuda.set_device(dev)
            kw["device_id"] = dev
        torch.distributed.init_process_group(backend or ("nccl" if cuda else "gloo"), timeout=dist_timeout(), **kw)
    return resolve()
