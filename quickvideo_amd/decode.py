"""Greedy decode over the pruned cache as ONE captured hipGraph per token (a10: the reference decodes through HF `generate`
with the LVU cache, lvu/models/qwen25_lvu.py:744-761, ~12 kernel launches per layer issued from Python).

A decode step is launch-bound when driven from the host (≈340 launches of a few µs each for 28 layers) while its real cost is
one pass over the weights (HBM-bound).  Here every kernel of the step reads its per-token scalars (cache rows in use, rotary
position) from a device state block, so the step is captured once (`torch.cuda.CUDAGraph` = hipGraph on ROCm) and replayed:

    embed(tok) -> [ gemv(RMSNorm + qkv + bias) -> M-RoPE + KV append + single-query attention (one MFMA kernel) -> combine
                    -> gemv(o_proj + residual) -> gemv(RMSNorm + gate/up + SwiGLU) -> gemv(down + residual) ] x L
               -> gemv(RMSNorm + lm_head) -> argmax -> tok ;  state += 1

6 launches per layer (+ one M-RoPE table per token), all libquickprefill.so kernels (qp_decode.hip); arithmetic and bf16 rounding points are those of the
eager `QuickPrefillEngine.decode_step` path (same RMSNorm summation order, one rounding per torch op of the reference), only the
fp32 accumulation order inside the matrix-vector products differs from hipBLASLt's.
"""
from __future__ import annotations

from typing import List, Optional

import torch


class GraphDecoder:
    @staticmethod
    def supported(eng) -> bool:
        """Single-GPU engines on the HIP library; the eager per-op path serves the rest (tensor / pipeline parallel decode,
        CPU test doubles, query-aware pruning during decode)."""
        return (eng.device.type == "cuda" and eng.tp_size == 1 and eng.pp_size == 1 and not eng.cfg.do_top_k_for_query
                and eng.D == 128 and eng.hq // eng.hkv <= 8 and hasattr(eng.ops, "gemv")
                and eng.spec.hidden % 8 == 0 and eng.li % 8 == 0 and max(eng.spec.hidden, eng.li) <= 32256)

    def __init__(self, eng):
        assert self.supported(eng)
        self.eng = eng
        dev, dt, s = eng.device, eng.dtype, eng.spec
        L = len(eng.w.layers)
        e = lambda *shape, dtype=dt: torch.empty(*shape, dtype=dtype, device=dev)
        # device-resident step scalars, one flat array so that a single kernel advances them:
        #   state[l] = {rows in layer l's cache, rotary position}, pos3 = the position on the three M-RoPE streams
        self._scalars = torch.zeros(2 * L + 3, dtype=torch.int64, device=dev)
        self.state = self._scalars[:2 * L].view(L, 2)
        self.pos3 = self._scalars[2 * L:].view(3, 1)
        self.tok = torch.zeros(1, dtype=torch.int64, device=dev)
        self.h = e(1, s.hidden)
        self.qkv = e((eng.hq + 2 * eng.hkv) * eng.D)
        self.att = e(eng.hq, eng.D)
        self.act = e(eng.li)
        self.logits = e(eng.w.lm_head.shape[0])
        self.ws = eng.ops.decode_attn_workspace(eng.hq, eng.hkv)
        self.graph: Optional[torch.cuda.CUDAGraph] = None

    # -------------------------------------------------------------------------------------------------------------------
    def _enqueue_step(self):
        eng, ops, s = self.eng, self.eng.ops, self.eng.spec
        w, D, hs = eng.w, eng.D, eng.arena.head_stride
        torch.index_select(w.embed, 0, self.tok, out=self.h)
        h = self.h.view(-1)
        cos, sin = ops.mrope_table(self.pos3, s.mrope_section, s.rope_theta, D)      # once per token, shared by all layers
        for l, lw in enumerate(w.layers):
            st = self.state[l]
            ops.gemv(lw.w_qkv, h, self.qkv, ops.GEMV_BIAS, bias=lw.b_qkv, norm_w=lw.ln1, eps=s.rms_eps)      # qwen25_lvu.py:167-169, 42-44
            ops.decode_attn_fused(self.qkv, cos, sin, st, eng.arena.k(l), eng.arena.v(l), hs, eng.hq, eng.hkv, D, D ** -0.5,
                                  self.att, self.ws)                         # M-RoPE + KV append + attention (:46-58, :61-112)
            ops.gemv(lw.w_o, self.att.view(-1), h, ops.GEMV_RESIDUAL)                                          # :114-115, :182
            ops.gemv(lw.w_gate_up, h, self.act, ops.GEMV_SWIGLU, norm_w=lw.ln2, eps=s.rms_eps)                # :195-197
            ops.gemv(lw.w_down, self.act, h, ops.GEMV_RESIDUAL)                                               # :197-198
        ops.gemv(w.lm_head, h, self.logits, ops.GEMV_BIAS, norm_w=w.norm, eps=s.rms_eps)
        torch.argmax(self.logits.float(), dim=0, keepdim=True, out=self.tok)
        ops.decode_advance(self._scalars)

    def _load_state(self, rope_delta: int):
        eng = self.eng
        p = eng.seq_pos + rope_delta
        vals = [x for n in eng.arena.len for x in (n, p)] + [p, p, p]
        self._scalars.copy_(torch.tensor(vals, dtype=torch.int64).to(self._scalars.device))

    def _capture(self, rope_delta: int):
        """One eager step warms every kernel (module load is not capturable), then the step is captured.  The warm-up writes the
        K/V row at `kv_len` and advances the state; the state is reloaded and the row is rewritten by the first real step."""
        side = torch.cuda.Stream(self.eng.device)
        side.wait_stream(torch.cuda.current_stream(self.eng.device))
        with torch.cuda.stream(side):
            self._enqueue_step()
        torch.cuda.current_stream(self.eng.device).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._enqueue_step()
        self.graph = g
        torch.cuda.synchronize(self.eng.device)

    # -------------------------------------------------------------------------------------------------------------------
    def begin(self, rope_delta: int):
        """(Re)bind to the engine's current cache state; call after the prompt tail and before the first `step`."""
        eng = self.eng
        if max(eng.arena.len) + 1 > eng.arena.capacity:
            raise RuntimeError(f"KV arena full ({max(eng.arena.len)} of {eng.arena.capacity} rows): no room to decode")
        if self.graph is None:
            keep = self.tok.clone()
            self._load_state(rope_delta)
            self._capture(rope_delta)
            self.tok.copy_(keep)
        self._load_state(rope_delta)

    def step(self, token: Optional[int] = None) -> torch.Tensor:
        """Decode one token: feeds `token` (or, when None, the argmax the previous step left on the device), returns the bf16
        logits buffer [V] (valid until the next step).  The greedy next token stays on the device in `self.tok`."""
        eng = self.eng
        if max(eng.arena.len) + 1 > eng.arena.capacity:
            raise RuntimeError(f"KV arena full ({eng.arena.capacity} rows)")
        if token is not None:
            self.tok.fill_(int(token))
        self.graph.replay()
        eng.arena.len = [n + 1 for n in eng.arena.len]
        eng.seq_pos += 1
        return self.logits

    def generate(self, first_token: int, max_new_tokens: int, rope_delta: int, eos_token_id=None) -> List[int]:
        """Greedy continuation after `first_token` (the TTFT token): up to max_new_tokens further tokens.  eos_token_id: an id or a
        collection of ids — HF generate stops on ANY id of generation_config.eos_token_id ([151645, 151643] for Qwen2/2.5-VL)."""
        if eos_token_id is not None and not isinstance(eos_token_id, (set, frozenset)):
            eos_token_id = frozenset(int(e) for e in (eos_token_id if isinstance(eos_token_id, (list, tuple)) else [eos_token_id]))
        if not eos_token_id:
            eos_token_id = None
        if max_new_tokens <= 0:
            return []
        self.begin(rope_delta)
        out: List[int] = []
        tok = int(first_token)
        self.tok.fill_(tok)
        if eos_token_id is None:                      # no stop condition: no host round trip per token either
            room = self.eng.arena.capacity - max(self.eng.arena.len)
            toks = torch.empty(min(max_new_tokens, room), dtype=torch.int64, device=self.eng.device)
            for i in range(toks.numel()):
                self.step()
                toks[i:i + 1].copy_(self.tok)
            return [int(t) for t in toks.tolist()]
        for _ in range(max_new_tokens):
            if tok in eos_token_id:
                break
            self.step()
            tok = int(self.tok.item())
            out.append(tok)
        return out
