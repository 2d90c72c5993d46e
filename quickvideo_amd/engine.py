"""QuickPrefill engine: group-chunked prefill with per-group key-norm KV pruning on one MI355X
(or one tensor-parallel rank).

Mirrors, per group, the reference's patched decoder layer (lvu/models/qwen25_lvu.py:122-212) and attention
(:29-120) with the pruning hook of lvu/utils.py:197-376 — but laid out for the hardware:

  * one pre-allocated KV arena [L][2][Hkv][capacity][D] bf16 for the whole video (the reference re-allocates
    and copies the whole past twice per layer per group: torch.cat in cache.update and in utils.py:335-336);
  * the group's new K/V are written by the fused RoPE kernel into a small staging block, attention reads
    (arena prefix, staging) as two segments, and pruning is a pure gather staging -> arena tail;
    layers that do not prune append straight into the arena;
  * no host round trip inside the loop (the reference syncs 2x per layer: utils.py:136, :284);
  * GQA is native in the attention kernel (no repeat_kv materialisation, :61-62);
  * lm_head only for the last position of the prompt tail (HF 4.50 computes it for every video token).

Two drivers of the same launches: `_forward_segment_native` hands a whole segment (all layers) to the library in ONE call
(qp_prefill_segment: single-device key-norm path; GEMMs through the library's hipBLASLt plans), `forward_segment` issues them
operator by operator (every other mode and layout; GEMMs through torch.mm or the library, whichever the warm-up tuner measured
faster).  Everything that is not a GEMM is a hand-written kernel of libquickprefill.so via quickvideo_amd.native.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from .lvu_config import LVUConfig, NORM_PRUNE_MODES, QUERY_PRUNE_MODES, effective_k
from .weights import DecoderWeights


_TUNE_FAILURES: dict = {}       # (projection, rows, weight shape, bias) -> message: every shape hipBLASLt plan selection failed for


class _NoNativePlan(Exception):
    """qp_linear_tune found no usable hipBLASLt candidate for one projection of the one-call path at this row count."""


_NO_NATIVE = "no-native-plan"


def _tune_failed(shape_key, exc) -> None:
    """qp_linear_tune threw for this projection shape: the engine stays on torch.mm there (a correct but possibly slower GEMM — a
    broken plan path would otherwise cost ~5 % of the pass and nobody would know).  Logged once per shape on stderr and kept in
    `engine._TUNE_FAILURES` (bench.py prints the table in its line)."""
    if shape_key in _TUNE_FAILURES:
        return
    _TUNE_FAILURES[shape_key] = f"{type(exc).__name__}: {exc}"
    import sys
    print(f"[quickprefill] hipBLASLt plan selection failed for {shape_key}: {type(exc).__name__}: {exc} — staying on torch.mm for this shape",
          file=sys.stderr, flush=True)


def sp_row_ranges(n: int, world: int, rank: int):
    """Group-token parallel row assignment: the n rows are cut into 2*world chunks of ceil(n / (2*world)) rows; rank r owns
    chunks r and 2*world-1-r, which balances the causal attention work.  Returns ((a0, a1), (b0, b1)), possibly empty ranges."""
    m2 = -(-n // (2 * world))
    a0, a1 = min(n, rank * m2), min(n, (rank + 1) * m2)
    b0, b1 = min(n, (2 * world - 1 - rank) * m2), min(n, (2 * world - rank) * m2)
    return (a0, a1), (b0, b1)


class KVArena:
    """Pre-allocated per-layer KV store.  k(l)/v(l): [Hkv_local, capacity, D]; len[l] = rows in use."""

    def __init__(self, n_layers: int, n_kv_heads: int, capacity: int, head_dim: int, device, dtype=torch.bfloat16):
        self.capacity, self.head_dim, self.n_kv = capacity, head_dim, n_kv_heads
        self.buf = torch.empty(n_layers, 2, n_kv_heads, capacity, head_dim, dtype=dtype, device=device)
        self.len: List[int] = [0] * n_layers

    def k(self, l): return self.buf[l, 0]
    def v(self, l): return self.buf[l, 1]

    @property
    def head_stride(self) -> int:
        return self.capacity * self.head_dim

    def reset(self):
        self.len = [0] * len(self.len)


class QuickPrefillEngine:
    _SHARED: dict = {}
    def __init__(self, weights: DecoderWeights, cfg: LVUConfig, capacity: int, max_group_tokens: int, device=None, ops=None,
                 tp_group=None, sp_group=None, sp_rank: int = 0, sp_size: int = 1, pp_group=None, pp_rank: int = 0, pp_size: int = 1,
                 pp_peers: Optional[List[int]] = None, pp_send_group=None, pp_recv_group=None):
        self.w, self.spec, self.cfg = weights, weights.spec, cfg
        self.device = torch.device(device if device is not None else weights.embed.device)
        if ops is None:
            from .native import QuickPrefillOps      # raises QuickPrefillUnavailable without a GPU / library: no fallback
            ops = QuickPrefillOps(self.device)
        self.ops = ops
        self.tp_group = tp_group
        self.tp_size = weights.tp_size
        self.tp_rank = weights.tp_rank
        # tensor parallel: the two row-parallel projections of a layer (o_proj, down_proj) are cut into `tp_chunks` row blocks and block i's
        # all-reduce is issued asynchronously while block i+1's GEMM runs (rows are independent in a GEMM and an all-reduce is
        # element-wise, so the result is the unsplit one — bit for bit on 1 and 2 ranks; on more ranks a ring all-reduce adds the ranks'
        # partials in an order that depends on where an element sits in the buffer, like any change of message size would).
        # QP_TP_CHUNKS=1 is the unsplit form (both all-reduces fully exposed on the compute stream) for A/B timing.
        self.tp_chunks = max(1, int(os.environ.get("QP_TP_CHUNKS", "2")))
        # group-token parallelism ("sp"): weights and the KV arena are replicated (288 GB of HBM per GPU), every rank takes a
        # contiguous slice of each group's tokens through all layers and the ranks exchange only the group's new K/V rows
        # (+ key sums) once per layer — ~14x fewer bytes on xGMI than the two [n, d] all-reduces of tensor parallelism.
        self.sp_group, self.sp_rank, self.sp_size = sp_group, sp_rank, sp_size
        # A process group handed in explicitly is USED even when it has one member: every collective of the layer loop is then
        # issued on it (all-reduce of [n, d], all-gather of the key sums / of the K|V|sums exchange block).  That is how a 1-GPU
        # box pre-flights the RCCL path (backend "nccl", world_size 1: dtype / shape / device-binding errors surface without a
        # second GPU — tests/test_gpu_engine.py, bench.py --nccl-preflight); with no group a single rank issues no collective.
        # (tp_on / sp_on are properties: bench.py attaches the groups after construction)
        assert not (self.sp_on and self.tp_on), "choose tensor parallel OR group-token parallel"
        # layer-pipeline parallelism ("pp"): rank r holds a contiguous slice of the layers (weights AND their KV), receives a group's
        # hidden rows from rank r-1, runs its layers and hands the rows to rank r+1.  No collective: one [n, d] point-to-point
        # hand-off per group and stage.  Groups flow through the stages back to back, so a video of G groups keeps
        # G / (G + N - 1) of the machine busy — the mode for long videos (cfg4: G = 450); short ones (cfg2: G = 4) use "sp".
        self.pp_group, self.pp_rank, self.pp_size = pp_group, pp_rank, pp_size
        # pp_peers[s] = global rank of this rank's counterpart in stage s.  A stage may itself be a group-token parallel ("sp") group:
        # its ranks own the same zigzag rows in every stage, so each hands its own rows to its counterpart (no re-shuffle).
        self.pp_peers = pp_peers
        # optional 2-rank process groups towards the next / from the previous stage (parallel.ParallelContext.pair_groups): one
        # communicator per direction, so a recv never queues behind this stage's own unmatched send
        self.pp_send_group, self.pp_recv_group = pp_send_group, pp_recv_group
        assert not (self.pp_size > 1 and self.tp_size > 1), "layer pipeline is not combined with tensor parallelism"
        self.l0, self.n_layers_total = weights.layer0, weights.n_layers_total
        s = self.spec
        self.hq, self.hkv, self.li = weights.local_q_heads, weights.local_kv_heads, weights.local_inter
        self.D = s.head_dim
        self.dtype = weights.embed.dtype
        self.n_max = max_group_tokens
        self.arena = KVArena(len(weights.layers), self.hkv, capacity, self.D, self.device, self.dtype)
        n, d, dev, dt = self.n_max, s.hidden, self.device, self.dtype
        e = lambda *shape, dtype=dt: torch.empty(*shape, dtype=dtype, device=dev)
        self.b_h, self.b_x, self.b_o, self.b_dn = e(n, d), e(n, d), e(n, d), e(n, d)
        self.b_qkv = e(n, (self.hq + 2 * self.hkv) * self.D)
        self.b_q, self.b_att = e(n, self.hq, self.D), e(n, self.hq, self.D)
        self.b_gu, self.b_act = e(n, 2 * self.li), e(n, self.li)
        self.b_stage = e(2, self.hkv, n + 2 * self.sp_size, self.D)
        self.b_xsend = self.b_xall = self.b_ss_all = None                                 # exchange buffers: _parallel_buffers()
        self.b_ss = e(self.hkv, n, dtype=torch.float32)
        self._parallel_buffers()
        self.b_idx = e(n, dtype=torch.int32)
        self.b_idx_pp = e(n, dtype=torch.int32) if self.pp_size > 1 else None   # original rows of a hidden-pruned hand-off
        # prune through 16-bit norm keys (qp_prune_keys): the keys of the group's tokens, written by the RoPE kernel
        self._keys_path = hasattr(self.ops, "prune_keys") and self.device.type == "cuda"
        if self._keys_path:
            self.b_keys = e(n, dtype=torch.int16)
        self._prune_probe = os.environ.get("QP_PRUNE_PROBE") == "1"
        if self._prune_probe:
            self._probe_key = e(8, dtype=torch.int16)
        self.b_h2 = e(n, d)
        # norm-based predict type (utils.py:117-136): which rows are scored (keys / values) and which end is kept
        self.query_mode = cfg.top_k_predict_type in QUERY_PRUNE_MODES     # query-attention-score pruning (lvu_cache.py:97-117)
        if cfg.top_k_predict_type not in NORM_PRUNE_MODES and not self.query_mode:
            raise ValueError(f"Unknown predict type: {cfg.top_k_predict_type} (the native engine implements the norm-based modes "
                             f"{sorted(NORM_PRUNE_MODES)} and the query-score modes {sorted(QUERY_PRUNE_MODES)}; lvu/utils.py:55-62, 117-136)")
        self.norm_source, self.norm_order = NORM_PRUNE_MODES.get(cfg.top_k_predict_type, (0, 1))
        if self.query_mode and not hasattr(self, "b_keys"):
            self.b_keys = e(n, dtype=torch.int16)
        # per-call argument of every mode-dependent operator (no state on the shared ops object: two engines of different
        # top_k_predict_type may share one QuickPrefillOps)
        self.prune_mode = (self.norm_source << 1) | self.norm_order
        env = os.environ.get("QP_SPLIT_GATE_UP_ROWS")                                  # developer override, see _gate_up_swiglu
        self.split_gate_up_rows = tuple(int(v) for v in env.split(",")) if env else None
        self._tune_gemms = self.device.type == "cuda" and os.environ.get("QP_TUNE_GEMMS", "1") == "1"
        # QP_GEMM_BACKEND=lt: EVERY projection goes through the library's own hipBLASLt path (qp_linear_act, bound without releasing the
        # interpreter lock) instead of torch.mm — no torch call is left in the layer loop, so another Python thread that wants the lock
        # gets it at the switch interval only, not at every one of the ~150 GEMM calls of a group (bench.py host_contention)
        self._lt_only = (self.device.type == "cuda" and os.environ.get("QP_GEMM_BACKEND", "") == "lt" and hasattr(self.ops, "linear_tune"))
        # decisions are per (projection shape, rows, device) and shared by every engine of the process
        self._gemm_plans, self._gu_split, self._lt_tuned = (QuickPrefillEngine._SHARED.setdefault((str(self.device), self._tune_gemms, i), {}) for i in range(3))
        # One call per segment instead of ~13 per layer: qp_prefill_segment sequences the same launches inside the library (single-device
        # key-norm path; everything else keeps the per-operator loop below).  QP_NATIVE_SEGMENT=0: the per-operator loop, for A/B.
        self._native = (self.device.type == "cuda" and hasattr(self.ops, "prefill_segment") and hasattr(self.ops, "linear_tune")
                        and os.environ.get("QP_NATIVE_SEGMENT", "1") == "1")
        self._native_state = None
        self.attn_timer = None                      # bench.py: .pairs(n_layers, what[, used]) -> (c_void_p * 2L) of hipEvent_t recorded around each attention / prune launch
        self.kept_trace: Optional[list] = None      # tests: set to [] to record kept indices per (group, layer)
        self.hidden_trace: Optional[list] = None    # tests: set to [] to record the residual stream after every layer (fp32 copy)
        self.seq_pos = 0                            # tokens of the original sequence consumed so far

    # ------------------------------------------------------------------ helpers
    @property
    def tp_on(self) -> bool:
        return self.tp_size > 1 or self.tp_group is not None

    @property
    def sp_on(self) -> bool:
        return self.sp_size > 1 or self.sp_group is not None

    def _parallel_buffers(self):
        """Exchange buffers of the parallel layouts, allocated when the layout is (or becomes) active."""
        n, dev = self.n_max, self.device
        if self.sp_on and self.b_xsend is None:
            m = 2 * -(-n // (2 * self.sp_size))                                           # two chunks of ceil(n / 2N) rows
            self.sp_chunk = 2 * self.hkv * m * self.D * 2 + self.hkv * m * 4              # bytes: K | V | key sums of one rank
            self.b_xsend = torch.empty(self.sp_chunk, dtype=torch.uint8, device=dev)
            self.b_xall = torch.empty(self.sp_size * self.sp_chunk, dtype=torch.uint8, device=dev)
        if self.tp_on and self.b_ss_all is None:
            self.b_ss_all = torch.empty(self.tp_size, self.hkv, n, dtype=torch.float32, device=dev)

    def reset(self):
        self.pp_flush()
        self.arena.reset()
        self.seq_pos = 0

    def _prune(self, ss_all, heads_total, n, k_keep, kn, vn, new_stride, l, past, idx, keys_ready=False):
        """post_process_kv_cache's KV part (utils.py:266-342): select the k_keep tokens + move their K/V rows staging -> arena tail."""
        ops, D = self.ops, self.D
        if not (self._keys_path and n <= ops.PRUNE_KEYS_MAX_N):
            ops.prune_staged(ss_all, heads_total, n, k_keep, kn, vn, new_stride, self.hkv, D, self.arena.k(l), self.arena.v(l),
                             self.arena.head_stride, past, idx, mode=self.prune_mode)
            return
        if not keys_ready:                                   # sums crossed ranks / came from the value rows: keys now
            ops.norm_keys(ss_all, heads_total, n, self.b_keys, mode=self.prune_mode)
        elif self._prune_probe:                              # developer probe: an (almost) empty launch in the prune's position
            ops.norm_keys(self.b_ss, 1, 1, self._probe_key)
        ops.prune_keys(self.b_keys, n, k_keep, kn, vn, new_stride, self.hkv, D, self.arena.k(l), self.arena.v(l), self.arena.head_stride,
                       past, idx)

    # ------------------------------------------------------------------ GEMM decomposition (hipBLASLt shape sensitivity)
    # hipBLASLt's heuristic is erratic in the row count: at 7B dims the down projection takes 576 us for n = 5760 rows but 940-950 us
    # for n = 5775..5824 (o_proj 127 vs 185 us), and one fused [n, 2I] gate/up GEMM is 4-7 % slower than two [n, I] GEMMs for
    # n = 2240..3600 but 7 % faster at 5760 (tools/bench_gemm.py, tools/probe/probe_gate_up_split.py).  So the first time a
    # segment size shows up (the warm-up step) every projection is timed in a few row decompositions — whole, or rows
    # [0, floor(n / q) * q) + the remainder for q in (64, ..., 4096) — and gate/up also as two GEMMs; the fastest is kept if it wins
    # by > 3 %.  Rows are independent in a GEMM, so every decomposition computes the same projection (fp32 accumulation order
    # may differ between kernels, like between any two hipBLASLt algorithms).
    def _run_linear(self, plan, x, w, out, bias):
        if plan == "lt":                              # hipBLASLt algorithm picked by qp_linear_tune for this problem
            self.ops.linear_act(x, w, bias, out, self.ops.ACT_NONE)
            return
        for r0, r1 in plan:
            if bias is None:
                torch.mm(x[r0:r1], w.t(), out=out[r0:r1])
            else:
                torch.addmm(bias, x[r0:r1], w.t(), out=out[r0:r1])

    def _time(self, fn) -> float:
        for _ in range(2):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(4):
            fn()
        e.record()
        e.synchronize()
        return s.elapsed_time(e) / 4

    def _row_plans(self, n: int):
        plans = [[(0, n)]]
        for q in (64, 128, 256, 512, 1024, 2048, 4096):
            m = n // q * q
            if 0 < m < n and [(0, m), (m, n)] not in plans:
                plans.append([(0, m), (m, n)])
        return plans

    def _small_linear(self, key: str, x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, bias=None) -> bool:
        """Segments of a few dozen rows (prompt tail, eager decode): the GEMM is a pass over the weights and hipBLASLt's default pick
        streams them at 1.2-2.4 TB/s when they are cold; the library times all of its candidates over every layer's copy of this
        projection once (qp_linear_tune) and qp_linear_act then runs the fastest (down projection at 30 rows: 110 -> 47 us)."""
        n = x.shape[0]
        if not (self._tune_gemms and n < 256 and hasattr(self.ops, "linear_tune") and x.is_contiguous() and w.is_contiguous()):
            return False
        lk = (key, n, tuple(w.shape), bias is not None)
        if lk not in self._lt_tuned:
            ws = [getattr(lw, self._WKEY[key]) for lw in self.w.layers]
            try:
                self.ops.linear_tune(x, ws, bias, out, self.ops.ACT_NONE)
                self._lt_tuned[lk] = True
            except Exception as e:                    # no usable candidate: stay on torch.mm for this shape — and SAY so, once per shape
                self._lt_tuned[lk] = False
                _tune_failed(lk, e)
        if not self._lt_tuned[lk]:
            return False
        self.ops.linear_act(x, w, bias, out, self.ops.ACT_NONE)
        return True

    _WKEY = {"qkv": "w_qkv", "o": "w_o", "gate_up": "w_gate_up", "down": "w_down"}

    def _lt_linear(self, key: str, x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, bias=None) -> bool:
        """hipBLASLt through the library for ANY row count (QP_GEMM_BACKEND=lt); False when plan selection failed for this shape."""
        lk = ("lt", x.shape[0], w.shape[0], w.shape[1], bias is not None)     # = the library's plan key: tuned once, whoever asks first
        ok = self._lt_tuned.get(lk)
        if ok is None:
            try:
                self.ops.linear_tune(x, [getattr(lw, self._WKEY[key]) for lw in self.w.layers], bias, out, self.ops.ACT_NONE)
                ok = True
            except Exception as e:
                ok = False
                _tune_failed(lk, e)
            self._lt_tuned[lk] = ok
        if ok:
            self.ops.linear_act(x, w, bias, out, self.ops.ACT_NONE)
        return ok

    def _linear(self, key: str, x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, bias=None):
        n = x.shape[0]
        if self._lt_only and x.is_contiguous() and w.is_contiguous() and out.is_contiguous() and self._lt_linear(key, x, w, out, bias):
            return
        if n < 256 and self._small_linear(key, x, w, out, bias):
            return
        pk = (key, n, tuple(w.shape), bias is not None)
        plan = self._gemm_plans.get(pk)
        if plan is None:
            plan = [(0, n)]
            if self._tune_gemms and n >= 256:
                best = None
                for cand in self._row_plans(n):
                    ms = self._time(lambda: self._run_linear(cand, x, w, out, bias))
                    if best is None:
                        best, whole = (ms, cand), ms
                    elif ms < best[0] and ms < 0.97 * whole:
                        best = (ms, cand)
                # ... and hipBLASLt's other heuristic candidates for the same problem (qp_linear_tune times them over every layer's
                # copy of the projection): torch.mm takes the library's FIRST candidate, which for the down projection at M = 2240 is
                # 357 us (0.85 PF) where another one runs 243 us (1.25 PF)  (tools/probe/probe_big_algos.py 2240)
                if (os.environ.get("QP_TUNE_LT", "1") == "1" and hasattr(self.ops, "linear_tune") and x.is_contiguous() and w.is_contiguous()
                        and out.is_contiguous()):
                    try:
                        ws = [getattr(lw, self._WKEY[key]) for lw in self.w.layers]
                        self.ops.linear_tune(x, ws, bias, out, self.ops.ACT_NONE)
                        ms = self._time(lambda: self.ops.linear_act(x, w, bias, out, self.ops.ACT_NONE))
                        if ms < best[0] and ms < 0.97 * whole:
                            best = (ms, "lt")
                    except Exception as e:            # no usable candidate for this shape: keep torch.mm — and SAY so, once per shape
                        _tune_failed(pk, e)
                plan = best[1]
                if os.environ.get("QP_ENGINE_DEBUG"):
                    print(f"[engine] {key} n={n}: whole {whole * 1e3:.0f} us -> {plan} {best[0] * 1e3:.0f} us", flush=True)
            self._gemm_plans[pk] = plan
        self._run_linear(plan, x, w, out, bias)

    def _gate_up_swiglu(self, x2: torch.Tensor, lw, act: torch.Tensor):
        """act = silu(x2 W_gate^T) * (x2 W_up^T) as one fused [n, 2I] GEMM or two [n, I] GEMMs, each whole or in the row decompositions
        of _row_plans, whichever hipBLASLt runs fastest for this row count (see above; QP_SPLIT_GATE_UP_ROWS="lo,hi" forces the
        whole-rows two-GEMM form for lo <= n < hi)."""
        n, li = x2.shape[0], self.li
        flat = self.b_gu.view(-1)
        g, u = flat[: n * li].view(n, li), flat[n * li: 2 * n * li].view(n, li)
        gu = self.b_gu[:n]

        def run(two, plan):
            for r0, r1 in plan:
                if two:
                    torch.mm(x2[r0:r1], lw.w_gate_up[:li].t(), out=g[r0:r1])
                    torch.mm(x2[r0:r1], lw.w_gate_up[li:].t(), out=u[r0:r1])
                else:
                    torch.mm(x2[r0:r1], lw.w_gate_up.t(), out=gu[r0:r1])

        if self._lt_only and x2.is_contiguous() and self._lt_linear("gate_up", x2, lw.w_gate_up, gu):
            self.ops.swiglu(gu, act)
            return
        if n < 256 and self._small_linear("gate_up", x2, lw.w_gate_up, gu):
            self.ops.swiglu(gu, act)
            return
        gk = (n, tuple(lw.w_gate_up.shape), self.split_gate_up_rows)
        choice = self._gu_split.get(gk)
        if choice is None:
            choice = (False, [(0, n)])
            if self.split_gate_up_rows is not None:
                choice = (self.split_gate_up_rows[0] <= n < self.split_gate_up_rows[1], [(0, n)])
            elif self._tune_gemms and n >= 256:
                whole = best = self._time(lambda: run(False, [(0, n)]))
                for two in (False, True):
                    for plan in self._row_plans(n):
                        if not two and plan == [(0, n)]:
                            continue
                        ms = self._time(lambda: run(two, plan))
                        if ms < best and ms < 0.97 * whole:
                            best, choice = ms, (two, plan)
                if os.environ.get("QP_ENGINE_DEBUG"):
                    print(f"[engine] gate_up n={n}: fused whole {whole * 1e3:.0f} us -> {'two GEMMs' if choice[0] else 'fused'} {choice[1]} "
                          f"{best * 1e3:.0f} us", flush=True)
            self._gu_split[gk] = choice
        two, plan = choice
        run(two, plan)
        if two:
            self.ops.swiglu_split(g, u, act)
        else:
            self.ops.swiglu(gu, act)

    def _all_reduce(self, t: torch.Tensor):
        if self.tp_on:
            torch.distributed.all_reduce(t, group=self.tp_group)

    def _linear_reduced(self, key: str, x: torch.Tensor, w: torch.Tensor, out: torch.Tensor) -> list:
        """Row-parallel projection + tensor-parallel all-reduce of `out` [n, d], overlapped: -> outstanding all-reduce handles (wait on them
        before `out` is read: _wait).  RCCL runs a collective on its own stream behind an event of the issuing stream, so all-reduce(block i)
        proceeds beside GEMM(block i+1); only the LAST block's all-reduce is exposed (1 / tp_chunks of the message)."""
        n = x.shape[0]
        chunks = self.tp_chunks if (self.tp_on and n >= 128 * self.tp_chunks) else 1
        if chunks == 1:
            self._linear(key, x, w, out)
            self._all_reduce(out)
            return []
        step = (-(-n // chunks) + 63) // 64 * 64
        works = []
        for r0 in range(0, n, step):
            r1 = min(n, r0 + step)
            self._linear(key, x[r0:r1], w, out[r0:r1])
            works.append(torch.distributed.all_reduce(out[r0:r1], group=self.tp_group, async_op=True))
        return works

    @staticmethod
    def _wait(works: list):
        for wk in works:
            wk.wait()                                # RCCL: the compute stream waits for the collective's event (no host block)

    def _global_sumsq(self, n: int) -> (torch.Tensor, int):
        """[Hkv_total, n] per-head sums in ascending head order, identical on every rank (partials are
        all-gathered and added in fixed head order so the norm is bit-stable across TP degrees)."""
        if not self.tp_on:
            return self.b_ss.view(-1)[: self.hkv * n].view(self.hkv, n), self.hkv
        local = self.b_ss.view(-1)[: self.hkv * n].view(self.hkv, n)
        allb = self.b_ss_all.view(-1)[: self.tp_size * self.hkv * n].view(self.tp_size * self.hkv, n)
        torch.distributed.all_gather_into_tensor(allb, local, group=self.tp_group)      # rank-major == ascending head order
        total = self.spec.n_kv_heads
        if total % self.tp_size == 0:
            return allb, total
        rep = self.tp_size // total                 # each KV head replicated on `rep` consecutive ranks (hkv_local == 1)
        return allb[::rep].contiguous(), total

    # ------------------------------------------------------------------ one segment through all layers
    def forward_segment(self, embeds: torch.Tensor, pos: torch.Tensor, prune: bool, video_group: bool = False,
                        row_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        """embeds [n, d] (device, engine dtype), pos int64 [3, n].  Returns the final hidden rows [n', d] (pre-norm)
        (group-token parallel segments: this rank's rows only).  row_idx (layer-pipeline stages behind a stage that pruned the hidden
        rows, prefill_prune_starting_layer): embeds holds rows `row_idx` of the n-token segment, ascending.  self._seg_rows is left
        as the original row indices of the returned rows (None = all n)."""
        s, ops, cfg, D = self.spec, self.ops, self.cfg, self.D
        n = pos.shape[1]                             # embeds may hold only this rank's rows (sp stage behind another stage)
        assert n <= self.n_max, f"group of {n} tokens exceeds max_group_tokens={self.n_max}"
        self._seg_rows = row_idx
        self._parallel_buffers()
        if self._sp_active(n, prune):
            return self._forward_segment_sp(embeds, pos, prune, video_group)
        if row_idx is not None:                      # positions of the surviving rows only (utils.py:344-372 gathers them the same way)
            pos = pos.index_select(1, row_idx.long())
            n = pos.shape[1]
        assert embeds.shape[0] == n
        if self._native_segment_ok(n, prune, row_idx) and self._native_gemm_plan(n) is not None:
            return self._forward_segment_native(embeds, pos, prune, video_group)
        L = self.n_layers_total                      # effective_k's decay uses the GLOBAL layer index / count
        cos, sin = ops.mrope_table(pos.contiguous(), s.mrope_section, s.rope_theta, D)
        hbufs, hsel = (self.b_h, self.b_h2), 0
        h = hbufs[0][:n]
        h.copy_(embeds)
        delta = None                                 # pending residual (MLP output of the previous layer)
        scale = D ** -0.5
        for l, lw in enumerate(self.w.layers):
            n = h.shape[0]
            x = self.b_x[:n]
            ops.add_rmsnorm(h, delta, lw.ln1, x, s.rms_eps)                  # h += delta; x = RMSNorm(h)   (qwen25_lvu.py:167-169)
            qkv = self.b_qkv[:n]
            self._linear("qkv", x, lw.w_qkv, qkv, lw.b_qkv)                  # q/k/v proj + bias             (:42-44)
            k_keep = effective_k(n, cfg, self.l0 + l, L) if prune else None  # utils.py:231-255
            if k_keep is not None and self.query_mode:
                # a query-score predict type outside a prompt-appended video group (do_top_k_for_query on the tail / decode): the
                # reference has no scores there and trips its own assertion (utils.py:56, 59)
                raise AssertionError("attn_weights_i should be 1D, but got None (query-based pruning needs the prompt-appended groups)")
            past = self.arena.len[l]
            assert past + (k_keep if k_keep is not None else n) <= self.arena.capacity, "KV arena overflow"
            q = self.b_q[:n]
            if k_keep is not None:                                           # prune layer: new K/V go to staging
                kn = self.b_stage[0].view(-1)[: self.hkv * n * D].view(self.hkv, n, D)
                vn = self.b_stage[1].view(-1)[: self.hkv * n * D].view(self.hkv, n, D)
                fuse = (self._keys_path and n <= ops.PRUNE_KEYS_MAX_N and not self.tp_on and self.norm_source == 0
                        and ops.can_fuse_keys(self.hq, self.hkv))
                if fuse:                                                     # 16-bit norm keys while the key rows are in registers
                    ops.rope_append_keys(qkv, cos, sin, self.hq, self.hkv, D, q, kn, vn, n * D, 0, None, self.b_keys, mode=self.prune_mode)
                else:
                    ops.rope_append(qkv, cos, sin, self.hq, self.hkv, D, q, kn, vn, n * D, 0, self.b_ss)
                new_stride = n * D
                if self.norm_source == 1:                                    # vector_norms*: score the value rows (utils.py:117-126)
                    ops.key_sumsq(vn, n * D, 0, n, self.hkv, D, self.b_ss)
                # tensor parallel: the key sums cross the ranks NOW (they only depend on the RoPE'd keys), so that this small all-gather
                # is not queued behind the o_proj all-reduces on the collective stream and the prune can run beside them
                ss_pre = self._global_sumsq(n) if not fuse else None
            else:                                                            # append in place                (:56-58)
                kc, vc = self.arena.k(l), self.arena.v(l)
                ops.rope_append(qkv, cos, sin, self.hq, self.hkv, D, q, kc, vc, self.arena.head_stride, past, None)
                kn, vn, new_stride = kc[:, past:], vc[:, past:], self.arena.head_stride
            att = self.b_att[:n]
            # adaptive_local_attention=False: VIDEO GROUPS are prefilled independently (no cross-group attention,
            # qwen25_lvu.py:700-714); their pruned K/V still accumulate in the arena, and the prompt tail / decode steps always
            # attend to all of it (:724-742), also when do_top_k_for_query prunes them.
            past_attn = 0 if (video_group and not cfg.adaptive_local_attention) else past
            ops.prefill_attn(q, self.arena.k(l), self.arena.v(l), self.arena.head_stride, past_attn, kn, vn, new_stride, n, self.hq,
                             self.hkv, D, scale, att)                        # :61-62, :102-112
            o = self.b_o[:n]
            o_works = self._linear_reduced("o", att.view(n, self.hq * D), lw.w_o, o)   # o_proj + all-reduce    (:114-115)
            prune_hidden = (k_keep is not None and cfg.enable and isinstance(cfg.prefill_prune_starting_layer, int)
                            and cfg.prefill_prune_starting_layer >= 0 and self.l0 + l >= cfg.prefill_prune_starting_layer)   # GLOBAL layer index
            if k_keep is not None:                                           # post_process_kv_cache         (:183-192)
                idx = self.b_idx[:k_keep]
                if fuse:
                    self._prune(None, 0, n, k_keep, kn, vn, n * D, l, past, idx, keys_ready=True)
                else:
                    ss_all, heads_total = ss_pre
                    self._prune(ss_all, heads_total, n, k_keep, kn, vn, n * D, l, past, idx)
                self.arena.len[l] = past + k_keep
                if self.kept_trace is not None:
                    self.kept_trace.append((l, idx.clone()))
            else:
                self.arena.len[l] = past + n
                if self.kept_trace is not None:
                    self.kept_trace.append((l, None))
            self._wait(o_works)                                              # (the prune above ran beside the last block's all-reduce)
            if prune_hidden:                                                 # utils.py:292-331, 344-372
                ops.add_inplace(h, o)
                hsel ^= 1
                hk = hbufs[hsel][:k_keep]
                ops.gather_rows(h, idx, k_keep, s.hidden * h.element_size(), hk)
                c2, s2 = torch.empty_like(cos[:k_keep]), torch.empty_like(sin[:k_keep])
                ops.gather_rows(cos, idx, k_keep, cos.shape[1] * cos.element_size(), c2)
                ops.gather_rows(sin, idx, k_keep, sin.shape[1] * sin.element_size(), s2)
                h, cos, sin, n = hk, c2, s2, k_keep
                if self.pp_size > 1:                                         # the next stage needs the survivors' ORIGINAL rows
                    self._seg_rows = idx.clone() if self._seg_rows is None else self._seg_rows.index_select(0, idx.long())
                x2 = self.b_x[:n]
                ops.add_rmsnorm(h, None, lw.ln2, x2, s.rms_eps)
            else:
                x2 = self.b_x[:n]
                ops.add_rmsnorm(h, o, lw.ln2, x2, s.rms_eps)                 # h += attn; x2 = RMSNorm(h)     (:182, :195-196)
            act = self.b_act[:n]
            self._gate_up_swiglu(x2, lw, act)                                # gate & up, act(gate) * up      (:197)
            dn = self.b_dn[:n]
            self._wait(self._linear_reduced("down", act, lw.w_down, dn))
            delta = dn
            if self.hidden_trace is not None:                                # the layer's output = h + MLP (the add itself is deferred)
                self.hidden_trace.append((l, h.float() + dn.float()))
        ops.add_inplace(h, delta)                                            # last residual                  (:198)
        return h

    # ------------------------------------------------------------------ one call per segment (qp_prefill_segment)
    def _native_segment_ok(self, n: int, prune: bool, row_idx) -> bool:
        if not self._native or self.tp_on or row_idx is not None or self.hidden_trace is not None or n < 1 or self._prune_probe:
            return False
        if not prune:
            return True
        if self.query_mode or self._hidden_prune_on(True) or self.norm_source != 0:
            return False
        return bool(self._keys_path and n <= self.ops.PRUNE_KEYS_MAX_N and self.ops.can_fuse_keys(self.hq, self.hkv))

    def _native_gemm_plan(self, n: int):
        """Row decomposition of the four projections for the one-call path, from timings of the library's OWN GEMM path (qp_linear_act
        after qp_linear_tune) at this row count: whole, or rows [0, q*floor(n/q)) + the remainder (hipBLASLt's pick is erratic in M, see
        _linear), gate/up fused or as two GEMMs.  -> (split_qkv, split_o, split_gate_up, split_down, gate_up_two)."""
        key = ("native", n, self.hq, self.hkv, self.li, self.spec.hidden)
        plan = self._gemm_plans.get(key)
        if plan is not None:
            return None if plan is _NO_NATIVE else plan
        try:
            plan = self._build_native_gemm_plan(n)
        except _NoNativePlan:
            plan = _NO_NATIVE                         # remembered: segments of n rows take the per-operator loop (torch.mm where the tuner failed)
        self._gemm_plans[key] = plan
        return None if plan is _NO_NATIVE else plan

    def _build_native_gemm_plan(self, n: int):
        lws, li, d = self.w.layers, self.li, self.spec.hidden
        x, act, att = self.b_x[:n], self.b_act[:n], self.b_att[:n].view(n, self.hq * self.D)

        def run(name, xin, wsel, out, bias, split):
            blocks = [(0, n)] if not split else [(0, split), (split, n)]
            for r0, r1 in blocks:
                xb, ob = xin[r0:r1], out[r0:r1]
                w0 = wsel(lws[0])
                lk = ("lt", r1 - r0, w0.shape[0], w0.shape[1], bias is not None)  # shared with _lt_linear: one tuning per GEMM shape
                if self._lt_tuned.get(lk) is None:
                    try:
                        self.ops.linear_tune(xb, [wsel(lw) for lw in lws], bias, ob, self.ops.ACT_NONE)
                        self._lt_tuned[lk] = True
                    except Exception as e:            # no usable hipBLASLt candidate for this shape: this row count stays on the per-operator loop
                        self._lt_tuned[lk] = False
                        _tune_failed(lk, e)
                if self._lt_tuned[lk] is not True:
                    raise _NoNativePlan(lk)
                self.ops.linear_act(xb, wsel(lws[0]), bias, ob, self.ops.ACT_NONE)

        def best_split(name, xin, wsel, out, bias):
            if not (self._tune_gemms and n >= 256):
                run(name, xin, wsel, out, bias, 0)
                return 0, 0.0
            cands = [0] + [p[0][1] for p in self._row_plans(n) if len(p) == 2]        # 0 = one GEMM; m = rows [0, m) + [m, n)
            times = [(self._time(lambda c=c: run(name, xin, wsel, out, bias, c)), c) for c in cands]
            whole = times[0][0]
            t, c = min(times)
            return (c, t) if t < 0.97 * whole else (0, whole)

        s_qkv, _ = best_split("qkv", x, lambda lw: lw.w_qkv, self.b_qkv[:n], lws[0].b_qkv)
        s_o, _ = best_split("o", att, lambda lw: lw.w_o, self.b_o[:n], None)
        s_dn, _ = best_split("down", act, lambda lw: lw.w_down, self.b_dn[:n], None)
        flat = self.b_gu.view(-1)
        gbuf, ubuf = flat[: n * li].view(n, li), flat[n * li: 2 * n * li].view(n, li)
        s_gu, t_f = best_split("gate_up", x, lambda lw: lw.w_gate_up, self.b_gu[:n], None)
        two = 0
        if self._tune_gemms and n >= 256:
            s_g, t_g = best_split("gate", x, lambda lw: lw.w_gate_up[:li], gbuf, None)
            if t_f > 0 and 2 * t_g < 0.97 * t_f:
                run("up", x, lambda lw: lw.w_gate_up[li:], ubuf, None, s_g)          # same shape as "gate": plans exist
                two, s_gu = 1, s_g
        plan = (s_qkv, s_o, s_gu, s_dn, two)
        if os.environ.get("QP_ENGINE_DEBUG"):
            print(f"[engine] one-call segment path, n={n}: row splits qkv/o/gate_up/down = {s_qkv}/{s_o}/{s_gu}/{s_dn}, gate_up as "
                  f"{'two GEMMs' if two else 'one GEMM'}", flush=True)
        return plan

    def _forward_segment_native(self, embeds: torch.Tensor, pos: torch.Tensor, prune: bool, video_group: bool) -> torch.Tensor:
        """forward_segment through qp_prefill_segment: the same launches in the same order, sequenced inside the library."""
        import ctypes
        from .native import QpLayer, QpSegment
        s, ops, cfg, D = self.spec, self.ops, self.cfg, self.D
        n, L, Lt = pos.shape[1], len(self.w.layers), self.n_layers_total
        st = self._native_state
        if st is None:
            layers = (QpLayer * L)()
            for l, lw in enumerate(self.w.layers):
                for f, t in (("ln1", lw.ln1), ("w_qkv", lw.w_qkv), ("b_qkv", lw.b_qkv), ("w_o", lw.w_o), ("ln2", lw.ln2), ("w_gate_up", lw.w_gate_up),
                             ("w_down", lw.w_down), ("k_cache", self.arena.k(l)), ("v_cache", self.arena.v(l))):
                    assert t.is_contiguous() or f in ("k_cache", "v_cache")
                    setattr(layers[l], f, t.data_ptr())
            seg = QpSegment()
            seg.n_layers, seg.hidden, seg.n_q_heads, seg.n_kv_heads, seg.head_dim, seg.intermediate = L, s.hidden, self.hq, self.hkv, D, self.li
            seg.rms_eps, seg.attn_scale, seg.cache_capacity = s.rms_eps, D ** -0.5, self.arena.capacity
            self.b_idx_all = torch.empty(L, self.n_max, dtype=torch.int32, device=self.device)
            for f, t in (("h", self.b_h), ("x", self.b_x), ("qkv", self.b_qkv), ("q", self.b_q), ("att", self.b_att), ("o", self.b_o), ("gate_up", self.b_gu),
                         ("act", self.b_act), ("down", self.b_dn), ("k_stage", self.b_stage[0]), ("v_stage", self.b_stage[1]),
                         ("norm_keys", getattr(self, "b_keys", None)), ("kept_idx", self.b_idx_all)):
                setattr(seg, f, None if t is None else t.data_ptr())
            seg.kept_idx_stride = self.n_max
            st = self._native_state = (seg, layers, (ctypes.c_int64 * L)(), (ctypes.c_int64 * L)())
        seg, layers, cache_len, k_keep = st
        cos, sin = ops.mrope_table(pos.contiguous(), s.mrope_section, s.rope_theta, D)
        keeps = [effective_k(n, cfg, self.l0 + l, Lt) if prune else None for l in range(L)]
        for l in range(L):
            cache_len[l], k_keep[l] = self.arena.len[l], (-1 if keeps[l] is None else keeps[l])
        seg.split_qkv, seg.split_o, seg.split_gate_up, seg.split_down, seg.gate_up_two_gemms = self._native_gemm_plan(n)
        h = self.b_h[:n]                             # (plan selection used x / att / act and the projection outputs as scratch, not h)
        h.copy_(embeds)
        seg.n, seg.prune_mode = n, self.prune_mode
        seg.attend_prefix = 0 if (video_group and not cfg.adaptive_local_attention) else 1
        seg.cos, seg.sin = cos.data_ptr(), sin.data_ptr()
        ws = ops.attn_workspace(n, self.arena.len if seg.attend_prefix else [0], self.hq, self.hkv)
        seg.attn_ws, seg.attn_ws_bytes = ws.data_ptr(), ws.numel()
        seg.attn_events = self.attn_timer.pairs(L, "attn") if self.attn_timer is not None else None
        seg.prune_events = self.attn_timer.pairs(L, "prune", [k is not None for k in keeps]) if self.attn_timer is not None else None
        ops.prefill_segment(seg, layers, cache_len, k_keep)
        for l in range(L):
            self.arena.len[l] = int(cache_len[l])
            if self.kept_trace is not None:
                self.kept_trace.append((l, None if keeps[l] is None else self.b_idx_all[l, :keeps[l]].clone()))
        self._seg_rows = None
        return h

    # ------------------------------------------------------------------ query-attention-score groups
    def _forward_segment_query(self, embeds: torch.Tensor, pos: torch.Tensor, m: int) -> torch.Tensor:
        """One video group in the reference's query-based mode (top_k_predict_type "query_attention_weights[_by_value_norm]"):
        the m prompt tokens are appended to the group's n tokens (qwen25_lvu.py:684-689), their K/V never enter the cache and their
        queries score the group's keys (LVUCache.update, lvu_cache.py:97-117); post_process_kv_cache keeps the k highest-scoring of
        the n group tokens (utils.py:55-62, 236-238).  Attention is what flash-attn computes for n+m queries over past+n keys with
        its bottom-right aligned causal mask (:102-112): query i sees keys j <= i + past - m — the first m queries see only part
        of the prefix, query i >= m sees the whole prefix and the group's keys up to i - m.  embeds [n+m, d], pos [3, n+m]."""
        s, ops, cfg, D = self.spec, self.ops, self.cfg, self.D
        nt = pos.shape[1]
        n = nt - m
        assert embeds.shape[0] == nt and n > 0 and nt <= self.n_max, f"group of {n}+{m} tokens exceeds max_group_tokens={self.n_max}"
        # (group-token parallel engines run these segments REPLICATED — every rank the whole group on its replica, see _sp_active — so
        # nothing of the sp exchange is involved here)
        if self.tp_on and (self.tp_size * self.hq != s.n_heads or not hasattr(ops, "query_head_sums")):
            raise NotImplementedError("query-attention-score pruning under tensor parallelism needs the q heads to divide over the ranks without "
                                      "padding (zero pad heads of a replicated kv head would enter the mean over heads)")
        if cfg.enable and isinstance(cfg.prefill_prune_starting_layer, int) and cfg.prefill_prune_starting_layer >= 0:
            raise NotImplementedError("query-attention-score pruning + hidden-state pruning: the reference drops the prompt rows there")
        if n > 32768:
            raise NotImplementedError(f"query-attention-score pruning: groups of at most 32768 tokens (one softmax row of qp_query_scores lives in LDS; got {n})")
        by_vnorm = QUERY_PRUNE_MODES[cfg.top_k_predict_type]
        L = self.n_layers_total
        cos, sin = ops.mrope_table(pos.contiguous(), s.mrope_section, s.rope_theta, D)
        h = self.b_h[:nt]
        h.copy_(embeds)
        delta, scale, hs = None, D ** -0.5, self.arena.head_stride
        for l, lw in enumerate(self.w.layers):
            x = self.b_x[:nt]
            ops.add_rmsnorm(h, delta, lw.ln1, x, s.rms_eps)
            qkv = self.b_qkv[:nt]
            self._linear("qkv", x, lw.w_qkv, qkv, lw.b_qkv)
            k_keep = effective_k(n, cfg, self.l0 + l, L)                     # utils.py:236-255: q_len -= prompt_length
            past = self.arena.len[l]
            q = self.b_q[:nt]
            if k_keep is not None:                                           # group + prompt K/V to staging; only the group's rows are read again
                kn = self.b_stage[0].view(-1)[: self.hkv * nt * D].view(self.hkv, nt, D)
                vn = self.b_stage[1].view(-1)[: self.hkv * nt * D].view(self.hkv, nt, D)
                stride = nt * D
                ops.rope_append(qkv, cos, sin, self.hq, self.hkv, D, q, kn, vn, stride, 0, None)
            else:                                                            # no pruning in this layer: the prompt rows land behind the group's
                assert past + nt <= self.arena.capacity, "KV arena overflow"  # and are overwritten by the next append
                kc, vc = self.arena.k(l), self.arena.v(l)
                ops.rope_append(qkv, cos, sin, self.hq, self.hkv, D, q, kc, vc, hs, past, None)
                kn, vn, stride = kc[:, past:], vc[:, past:], hs
            assert past + (k_keep if k_keep is not None else n) <= self.arena.capacity, "KV arena overflow"
            att = self.b_att[:nt]
            pa = past if cfg.adaptive_local_attention else 0
            # queries m .. n+m-1: whole prefix + the group's keys j <= i - m  == the standard launch on the query rows shifted by m
            ops.prefill_attn(q[m:], self.arena.k(l), self.arena.v(l), hs, pa, kn, vn, stride, n, self.hq, self.hkv, D, scale, att[m:])
            # queries 0 .. m-1: prefix keys j <= pa - m + i only (none at all for i < m - pa)
            z = max(0, m - pa)
            if z:
                att[:z].zero_()
            if m > z:
                ops.prefill_attn(q[z:m], None, None, hs, 0, self.arena.k(l), self.arena.v(l), hs, pa, self.hq, self.hkv, D, scale, att[z:m],
                                 q_row0=pa - (m - z), nq=m - z)
            o = self.b_o[:nt]
            o_works = self._linear_reduced("o", att.view(nt, self.hq * D), lw.w_o, o)
            if k_keep is not None:
                vss = None
                if by_vnorm:
                    ops.key_sumsq(vn, stride, 0, n, self.hkv, D, self.b_ss)
                    vss = self.b_ss
                if self.tp_on:
                    # heads are sharded: every rank sums its heads' probabilities over the prompt queries (bf16 [hq_local, n]); the blocks
                    # are all-gathered in rank = ascending head order and every rank takes the SAME mean over all heads (the single-device
                    # kernel is this composition), so the kept list is identical everywhere and each rank compacts its own kv heads
                    if getattr(self, "b_hs", None) is None:
                        self.b_hs = torch.empty(self.hq, self.n_max, dtype=torch.int16, device=self.device)
                        self.b_hs_all = torch.empty(self.tp_size * self.hq, self.n_max, dtype=torch.int16, device=self.device)
                    hs_loc = self.b_hs.view(-1)[: self.hq * n].view(self.hq, n)
                    hs_all = self.b_hs_all.view(-1)[: self.tp_size * self.hq * n].view(self.tp_size * self.hq, n)
                    ops.query_head_sums(q[n:], kn, stride, n, self.hq, self.hkv, D, hs_loc)
                    torch.distributed.all_gather_into_tensor(hs_all.view(torch.bfloat16), hs_loc.view(torch.bfloat16), group=self.tp_group)   # (16-bit patterns)
                    vss_all, kv_total = self._global_sumsq(n) if by_vnorm else (None, 0)
                    ops.query_scores_from_head_sums(hs_all, self.tp_size * self.hq, n, self.b_keys, value_sumsq=vss_all, n_kv_total=kv_total)
                else:
                    ops.query_scores(q[n:], kn, stride, n, self.hq, self.hkv, D, self.b_keys, value_sumsq=vss)
                idx = self.b_idx[:k_keep]
                if n <= ops.PRUNE_KEYS_MAX_N:
                    ops.prune_keys(self.b_keys, n, k_keep, kn, vn, stride, self.hkv, D, self.arena.k(l), self.arena.v(l), hs, past, idx)
                else:                                                        # large group: select on the ready-made keys + gather
                    ops.select_keys(self.b_keys, n, k_keep, idx)
                    ops.gather_kv(kn, vn, stride, idx, k_keep, self.hkv, D, self.arena.k(l), self.arena.v(l), hs, past)
                self.arena.len[l] = past + k_keep
                if self.kept_trace is not None:
                    self.kept_trace.append((l, idx.clone()))
            else:
                self.arena.len[l] = past + n
                if self.kept_trace is not None:
                    self.kept_trace.append((l, None))
            self._wait(o_works)
            x2 = self.b_x[:nt]
            ops.add_rmsnorm(h, o, lw.ln2, x2, s.rms_eps)
            act = self.b_act[:nt]
            self._gate_up_swiglu(x2, lw, act)
            dn = self.b_dn[:nt]
            self._wait(self._linear_reduced("down", act, lw.w_down, dn))
            delta = dn
        ops.add_inplace(h, delta)
        return h

    # ------------------------------------------------------------------ group-token parallel variant of forward_segment
    def _forward_segment_sp(self, embeds: torch.Tensor, pos: torch.Tensor, prune: bool, video_group: bool = False) -> torch.Tensor:
        """Rank r runs its token rows of the segment through every layer; per layer ONE all-gather moves the ranks' new K/V rows
        and key sums, after which every rank holds the whole group's K/V in its staging block, attends its own query rows
        (qp_prefill_attn_rows) and applies the identical prune to its arena replica.  Rows are dealt "zigzag" (sp_row_ranges):
        the segment is cut into 2N chunks and rank r takes chunks r and 2N-1-r, so every rank gets the same share of the causal
        attention work (an early, cheap chunk plus a late, expensive one).
        Returns this rank's hidden rows (callers only need them for the replicated prompt tail)."""
        s, ops, cfg, D, N, r = self.spec, self.ops, self.cfg, self.D, self.sp_size, self.sp_rank
        n = pos.shape[1]
        assert not self._hidden_prune_on(prune), "segments with hidden-state pruning run replicated under group-token parallelism (_sp_active)"
        (a0, a1), (b0, b1) = sp_row_ranges(n, N, r)
        m2 = -(-n // (2 * N))
        m = 2 * m2                                   # rows per rank slot in the exchange buffers
        nA, nB = a1 - a0, b1 - b0
        ml = nA + nB                                 # local rows: [chunk r | chunk 2N-1-r]; nA == m2 whenever nB > 0
        L = self.n_layers_total                      # effective_k's decay uses the GLOBAL layer index / count
        rows = torch.cat([torch.arange(a0, a1, device=self.device), torch.arange(b0, b1, device=self.device)])
        cos, sin = ops.mrope_table(pos.index_select(1, rows).contiguous(), s.mrope_section, s.rope_theta, D)
        h = self.b_h[:ml]
        if self.pp_size > 1 and self.pp_rank > 0:    # rows handed over by the same sp rank of the previous pipeline stage
            assert embeds.shape[0] == ml
            h.copy_(embeds)
        else:
            h[:nA].copy_(embeds[a0:a1])
            h[nA:].copy_(embeds[b0:b1])
        delta = None
        scale = D ** -0.5
        kv_bytes = self.hkv * m * D * 2
        send_k = self.b_xsend[:kv_bytes].view(self.dtype).view(self.hkv, m, D)
        send_v = self.b_xsend[kv_bytes:2 * kv_bytes].view(self.dtype).view(self.hkv, m, D)
        send_ss = self.b_xsend[2 * kv_bytes:2 * kv_bytes + self.hkv * m * 4].view(torch.float32).view(self.hkv, m)
        chunk = 2 * kv_bytes + self.hkv * m * 4
        stage = self.b_stage.view(-1)[: 2 * self.hkv * N * m * D].view(2, self.hkv, N * m, D)
        kn, vn, new_stride = stage[0], stage[1], N * m * D
        ss_loc = self.b_ss.view(-1)[: self.hkv * ml].view(self.hkv, ml)
        for l, lw in enumerate(self.w.layers):
            x = self.b_x[:ml]
            ops.add_rmsnorm(h, delta, lw.ln1, x, s.rms_eps)
            qkv = self.b_qkv[:ml]
            self._linear("qkv", x, lw.w_qkv, qkv, lw.b_qkv)
            k_keep = effective_k(n, cfg, self.l0 + l, L) if prune else None
            past = self.arena.len[l]
            assert past + (k_keep if k_keep is not None else n) <= self.arena.capacity, "KV arena overflow"
            q = self.b_q[:ml]
            if ml == m:                                                      # sums straight into the send block
                ops.rope_append(qkv, cos, sin, self.hq, self.hkv, D, q, send_k, send_v, m * D, 0, send_ss)
            else:
                ops.rope_append(qkv, cos, sin, self.hq, self.hkv, D, q, send_k, send_v, m * D, 0, ss_loc)
                send_ss[:, :ml].copy_(ss_loc)
            torch.distributed.all_gather_into_tensor(self.b_xall[: N * chunk], self.b_xsend[:chunk], group=self.sp_group)
            # [rank][K | V | sums] with each rank's two zigzag chunks -> staging block + key sums in token order (one launch)
            ss_all = self.b_ss.view(-1)[: self.hkv * n].view(self.hkv, n)
            ops.sp_unpack(self.b_xall, N, self.hkv, m2, D, n, kn, vn, new_stride, ss_all)
            if self.norm_source == 1 and k_keep is not None:                 # vector_norms*: every rank scores the gathered value rows
                ops.key_sumsq(vn, new_stride, 0, n, self.hkv, D, ss_all)
            att = self.b_att[:ml]
            past_attn = 0 if (video_group and not cfg.adaptive_local_attention) else past
            for (q0, lo_, hi_) in ((a0, 0, nA), (b0, nA, ml)):               # the two row chunks of this rank
                if hi_ > lo_:
                    ops.prefill_attn(q[lo_:hi_], self.arena.k(l), self.arena.v(l), self.arena.head_stride, past_attn, kn, vn, new_stride,
                                     n, self.hq, self.hkv, D, scale, att[lo_:hi_], q_row0=q0, nq=hi_ - lo_)
            o = self.b_o[:ml]
            self._linear("o", att.view(ml, self.hq * D), lw.w_o, o)
            if k_keep is not None:
                idx = self.b_idx[:k_keep]
                self._prune(ss_all, self.hkv, n, k_keep, kn, vn, new_stride, l, past, idx)
                self.arena.len[l] = past + k_keep
                if self.kept_trace is not None:
                    self.kept_trace.append((l, idx.clone()))
            else:
                self.arena.k(l)[:, past:past + n].copy_(kn[:, :n])
                self.arena.v(l)[:, past:past + n].copy_(vn[:, :n])
                self.arena.len[l] = past + n
                if self.kept_trace is not None:
                    self.kept_trace.append((l, None))
            x2 = self.b_x[:ml]
            ops.add_rmsnorm(h, o, lw.ln2, x2, s.rms_eps)
            act = self.b_act[:ml]
            self._gate_up_swiglu(x2, lw, act)
            dn = self.b_dn[:ml]
            self._linear("down", act, lw.w_down, dn)
            delta = dn
        ops.add_inplace(h, delta)
        return h

    # ------------------------------------------------------------------ public steps of the group loop
    # layer-pipeline hand-off: stage r > 0 receives the segment's hidden rows from stage r-1, the last stage keeps its output
    def _sp_active(self, n: int, prune: bool = False) -> bool:
        """Is this segment's work split over the sp ranks?  Short segments (prompt tail, decode) and segments that prune the HIDDEN rows
        (prefill_prune_starting_layer: the surviving rows of a layer are an irregular subset, which the fixed zigzag deal of the K/V
        exchange cannot follow), and every segment of a query-attention-score engine (the appended prompt rows score ALL of the group's keys),
        run REPLICATED instead: every rank computes the whole segment on its replica of the weights and the
        cache — same result on every rank, no exchange, no speed-up for that segment."""
        return self.sp_on and n >= 64 * self.sp_size and not self._hidden_prune_on(prune) and not (self.query_mode and self.cfg.enable)

    def _pp_rows(self, n: int, prune: bool = False) -> int:
        """Rows of an n-token segment that travel between this rank and its pipeline counterparts."""
        if not self._sp_active(n, prune):
            return n
        (a0, a1), (b0, b1) = sp_row_ranges(n, self.sp_size, self.sp_rank)
        return (a1 - a0) + (b1 - b0)

    def _pp_p2p_group(self, sending: Optional[bool] = None):
        pair = self.pp_send_group if sending else (self.pp_recv_group if sending is False else None)
        if pair is not None:
            return pair
        return None if self.pp_peers is not None else self.pp_group      # explicit peers are global ranks of the default group

    def _pp_peer(self, r: int) -> int:
        if self.pp_peers is not None:
            return self.pp_peers[r]
        return torch.distributed.get_global_rank(self.pp_group, r) if self.pp_group is not None else r

    def _pp_host_staged(self, t: torch.Tensor) -> bool:
        # gloo moves device tensors without ordering against the compute stream: stage through the host (developer runs of
        # several ranks on one GPU; RCCL point-to-point is stream-ordered and takes the device buffer directly)
        return t.is_cuda and torch.distributed.get_backend(self._pp_p2p_group()) == "gloo"

    def _hidden_prune_on(self, prune: bool) -> bool:
        p = self.cfg.prefill_prune_starting_layer
        return bool(prune and self.cfg.enable and isinstance(p, int) and p >= 0)

    def _rows_entering_stage(self, n: int, prune: bool):
        """Hidden rows of an n-token segment that reach this stage's first layer: the earlier stages' hidden-state prunes
        (prefill_prune_starting_layer, utils.py:292-331) are a function of (n, config, layer index) alone, so both ends of a
        hand-off agree on the row count without a message.  -> (rows, whether any earlier layer pruned the hidden rows)."""
        if not self._hidden_prune_on(prune):
            return n, False
        cur, pruned, L, pps = n, False, self.n_layers_total, self.cfg.prefill_prune_starting_layer
        for l in range(self.l0):
            k = effective_k(cur, self.cfg, l, L)
            if k is not None and l >= pps:
                cur, pruned = k, True
        return cur, pruned

    def _pp_recv(self, buf: torch.Tensor):
        src, grp = self._pp_peer(self.pp_rank - 1), self._pp_p2p_group(sending=False)
        if self._pp_host_staged(buf):
            host = torch.empty(buf.shape, dtype=buf.dtype)
            torch.distributed.recv(host, src=src, group=grp)
            buf.copy_(host)
        else:
            torch.distributed.recv(buf, src=src, group=grp)

    def _pp_in(self, embeds: torch.Tensor, n: Optional[int] = None, prune: bool = False):
        """-> (hidden rows entering this stage, their original row indices or None)."""
        if self.pp_size == 1 or self.pp_rank == 0:
            return embeds, None
        n = embeds.shape[0] if n is None else n
        rows, pruned = self._rows_entering_stage(n, prune)
        if not pruned:
            buf = self.b_h2[: self._pp_rows(n, prune)]
            self._pp_recv(buf)
            return buf, None
        buf, idx = self.b_h2[:rows], self.b_idx_pp[:rows]           # a stage before this one pruned the hidden rows
        self._pp_recv(buf)
        self._pp_recv(idx)
        return buf, idx

    def _pp_out(self, h: torch.Tensor):
        """Hand the segment's hidden rows (+ the survivors' original row indices after a hidden-state prune) to the next stage WITHOUT
        stalling this stage's compute stream: the rows are copied into one of two send slots and sent asynchronously (`isend`: RCCL
        runs point-to-point on its own stream behind an event of this one), so the stage goes straight on to its next segment while
        the transfer waits for the receiver to post its recv.  A slot is reused only after ITS send has completed (`wait()` = a stream
        wait under RCCL): a receiver that falls two segments behind holds the sender there — back-pressure, not a clobbered buffer.
        (The blocking `send` of rounds 1-3 made every stage wait for its successor to reach the matching recv before it could
        start its own next segment: lock-step instead of a pipeline whenever stage times jitter.)"""
        if not (self.pp_size > 1 and self.pp_rank < self.pp_size - 1):
            return
        dist = torch.distributed
        dst, grp = self._pp_peer(self.pp_rank + 1), self._pp_p2p_group(sending=True)
        if not hasattr(self, "_pp_slots"):
            self._pp_slots, self._pp_turn = [None, None], 0
            self._pp_sendbuf = torch.empty(2, self.n_max, self.spec.hidden, dtype=self.dtype, device=self.device)
            self._pp_sendidx = torch.empty(2, self.n_max, dtype=torch.int32, device=self.device)
        slot = self._pp_turn & 1
        self._pp_turn += 1
        self._pp_wait_slot(slot)
        m = h.shape[0]
        buf = self._pp_sendbuf[slot, :m]
        buf.copy_(h)
        payload = [buf.cpu() if self._pp_host_staged(buf) else buf]
        rows = getattr(self, "_seg_rows", None)
        if rows is not None:
            ib = self._pp_sendidx[slot, :rows.shape[0]]
            ib.copy_(rows.to(torch.int32))
            payload.append(ib.cpu() if self._pp_host_staged(ib) else ib)
        works = [dist.isend(t, dst=dst, group=grp) for t in payload]           # same order as the receiver's recv calls
        self._pp_slots[slot] = (works, payload)                                # (payload kept alive until the send has completed)

    def _pp_wait_slot(self, slot: int):
        pending = self._pp_slots[slot]
        if pending is not None:
            for wk in pending[0]:
                wk.wait()
            self._pp_slots[slot] = None

    def pp_flush(self):
        """Wait for every outstanding hand-off of this stage (end of a video; before the process group goes away)."""
        if hasattr(self, "_pp_slots"):
            self._pp_wait_slot(0)
            self._pp_wait_slot(1)

    @property
    def is_last_stage(self) -> bool:
        return self.pp_rank == self.pp_size - 1

    def prefill_group(self, embeds: torch.Tensor, pos: torch.Tensor, prompt_embeds: Optional[torch.Tensor] = None):
        """One video group (qwen25_lvu.py:671-717): KV appended + pruned; hidden output is discarded like the
        reference discards the group's logits (:697-699).  (Layer pipeline: `embeds` only matters on stage 0.)
        Query-based predict types: pass the prompt's embedding rows [m, d] and positions for n+m tokens (the group's and the NEXT m
        of the sequence, :684-689); only the group's n tokens count towards the sequence position."""
        if self.query_mode and self.cfg.enable:
            if prompt_embeds is None:
                raise ValueError("query-based top_k_predict_type: prefill_group needs the prompt embeddings (qwen25_lvu.py:684-686)")
            m = prompt_embeds.shape[0]
            assert pos.shape[1] == embeds.shape[0] + m, "positions for the group's tokens AND the appended prompt tokens are required"
            # layer pipeline: the n + m rows (group + appended prompt) travel from stage to stage like any segment
            x, _ = self._pp_in(torch.cat([embeds, prompt_embeds], 0), pos.shape[1], prune=False)
            h = self._forward_segment_query(x, pos, m)
            self._pp_out(h)
            self.seq_pos += embeds.shape[0]
            return
        x, rows = self._pp_in(embeds, pos.shape[1], prune=True)
        h = self.forward_segment(x, pos, prune=True, video_group=True, row_idx=rows)
        self._pp_out(h)
        self.seq_pos += embeds.shape[0]

    def prefill_tail(self, embeds: torch.Tensor, pos: torch.Tensor) -> Optional[torch.Tensor]:
        """Prompt tail over the pruned cache, no pruning (qwen25_lvu.py:724-742, enable = do_top_k_for_query).
        Returns fp32 logits [V] of the last position = distribution of the first generated token (TTFT point); None on
        layer-pipeline stages other than the last."""
        pr = bool(self.cfg.do_top_k_for_query)
        x, rows = self._pp_in(embeds, pos.shape[1], prune=pr)
        h = self.forward_segment(x, pos, prune=pr, row_idx=rows)
        self._pp_out(h)
        self.pp_flush()                              # the caller is about to read a result: nothing of this stage stays in flight
        self.seq_pos += embeds.shape[0]
        return self.logits_last(h) if self.is_last_stage else None

    def logits_last(self, h: torch.Tensor) -> torch.Tensor:
        if self.device.type == "cuda" and hasattr(self.ops, "gemv") and self.tp_size == 1 and self.spec.hidden <= 32256:   # (lm_head is not sharded by a 1-rank group)
            # final RMSNorm + lm_head of ONE row: the weight-streaming kernel of the decode step (1.09 GB at 6.5 TB/s)
            logits = torch.empty(self.w.lm_head.shape[0], dtype=self.dtype, device=self.device)
            self.ops.gemv(self.w.lm_head, h[-1].contiguous(), logits, self.ops.GEMV_BIAS, norm_w=self.w.norm, eps=self.spec.rms_eps)
            return logits.float()
        x = torch.empty(1, self.spec.hidden, dtype=self.dtype, device=self.device)
        self.ops.add_rmsnorm(h[-1:].contiguous(), None, self.w.norm, x, self.spec.rms_eps)
        return torch.mm(x, self.w.lm_head.t())[0].float()

    def decode_step(self, token_embed: torch.Tensor, rope_delta: int) -> torch.Tensor:
        """One greedy decode step: position = original sequence position + rope_delta on all three streams
        (HF generate with the caller-supplied cache_position, qwen25_lvu.py:445-464, 740)."""
        p = self.seq_pos + rope_delta
        pos = torch.full((3, 1), p, dtype=torch.int64, device=self.device)
        x, rows = self._pp_in(token_embed.view(1, -1), 1, prune=bool(self.cfg.do_top_k_for_query))
        h = self.forward_segment(x, pos, prune=bool(self.cfg.do_top_k_for_query), row_idx=rows)
        self._pp_out(h)
        self.pp_flush()
        self.seq_pos += 1
        return self.logits_last(h) if self.is_last_stage else None

    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        return self.w.embed.index_select(0, ids.to(self.device))
