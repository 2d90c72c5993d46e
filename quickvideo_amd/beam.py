"""Beam search over the decode leg (a10): the reference forwards `num_beams` (and every other generation kwarg) to HF `generate`
(lvu/models/qwen25_lvu.py:740-761), so a drop-in has to offer it.  This restates transformers' beam search (generation/utils.py
`_beam_search` [3P], the vectorised form of 4.50+/5.x) for batch size 1, deterministic (`do_sample=False`), one returned sequence:

  every step    log-softmax of each running beam's logits (+ the repetition-penalty processor on the beam's own tokens) + the beam's
                accumulated score -> the K = max(2, 1 + #eos) * B best (beam, token) continuations over all B * V candidates
  running beams the B best continuations that did NOT just end (eos / length limit)
  finished      continuations among the top B that ended: score / (generated length ** length_penalty), merged into the B best finished
  stop          no running beam can beat the worst finished one any more (heuristic of `early_stopping=False`: the best running score
                at the current length), or — `early_stopping=True` — B finished sequences exist, or nothing can continue (max_new_tokens)

The model is a callback `advance(parents, tokens) -> fp32 logits [B, V]`: running beam i of the next step extends beam `parents[i]` of
this step by `tokens[i]`.  The engine side (`EngineBeams`) keeps ONE KV arena: the prefilled video + prompt rows are shared by all beams
(504 k rows per layer for the 1-hour video), each beam only owns the few rows of its generated tokens, which are swapped into the
arena tail before the beam's decode step (B sequential steps per token — HF batches them; the answer is the same).
Pinned against the installed transformers' `generate(num_beams=...)` on a tiny text model (tests/test_beam_search.py)."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

NEG = -1.0e9


def beam_search(first_logits: torch.Tensor, advance: Callable[[List[int], List[int]], torch.Tensor], num_beams: int, max_new_tokens: int,
                eos_ids: Sequence[int] = (), length_penalty: float = 1.0, early_stopping=False, repetition_penalty: float = 1.0,
                prompt_ids: Sequence[int] = (), do_sample: bool = False, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0,
                generator: Optional[torch.Generator] = None) -> List[int]:
    """-> generated token ids of the best hypothesis (eos included when it ended the sequence).  first_logits: fp32 [V] logits of the
    prompt's last position (every beam starts from it: HF initialises the scores as [0, -1e9, ...] so that step 0 spreads ONE beam).
    do_sample: beam-search multinomial sampling — the warpers (temperature, top-k, top-p) act on each beam's log-probabilities and the K
    continuations are DRAWN without replacement from softmax(accumulated scores) instead of taken by top-K (HF `_get_top_k_continuations`)."""
    B, V = int(num_beams), first_logits.numel()
    if B < 1 or max_new_tokens < 1:
        raise ValueError("num_beams and max_new_tokens must be >= 1")
    eos = sorted(set(int(e) for e in eos_ids))
    K = max(2, 1 + len(eos)) * B
    dev = first_logits.device
    seqs: List[List[int]] = [[] for _ in range(B)]                  # generated tokens of the running beams
    run_scores = torch.full((B,), NEG, dtype=torch.float32, device=dev)
    run_scores[0] = 0.0
    fin_seqs: List[List[int]] = [[] for _ in range(B)]
    fin_scores = torch.full((B,), NEG, dtype=torch.float32, device=dev)
    fin_done = [False] * B
    improvable = True
    logits = first_logits.float().unsqueeze(0).expand(B, V)
    eos_t = torch.tensor(eos, dtype=torch.long, device=dev) if eos else None
    seen0 = None
    if repetition_penalty != 1.0:
        seen0 = torch.zeros(V, dtype=torch.bool, device=dev)
        if len(prompt_ids):
            seen0[torch.as_tensor([p for p in prompt_ids if p < V], dtype=torch.long, device=dev)] = True
    for step in range(max_new_tokens):
        logp = torch.log_softmax(logits.float(), dim=-1)
        if repetition_penalty != 1.0:                               # RepetitionPenaltyLogitsProcessor on the log-probs, per beam
            logp = logp.clone()
            for b in range(B):
                seen = seen0.clone()
                if seqs[b]:
                    seen[torch.as_tensor(seqs[b], dtype=torch.long, device=dev)] = True
                row = logp[b]
                logp[b] = torch.where(seen, torch.where(row < 0, row * repetition_penalty, row / repetition_penalty), row)
        if do_sample:                                               # TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper per beam
            logp = logp.clone()
            if temperature != 1.0:
                logp = logp / temperature
            if top_k > 0:
                kth = torch.topk(logp, min(top_k, V), dim=-1).values[:, -1:]
                logp = logp.masked_fill(logp < kth, float("-inf"))
            if top_p < 1.0:
                srt, order = torch.sort(logp, descending=False, dim=-1)
                remove = srt.softmax(-1).cumsum(-1) <= (1.0 - top_p)
                remove[:, -1:] = False
                logp = logp.masked_fill(torch.zeros_like(remove).scatter(1, order, remove), float("-inf"))
        acc = (logp + run_scores[:, None]).reshape(-1)
        if do_sample:
            top_i = torch.multinomial(torch.softmax(acc, dim=-1), num_samples=K, generator=generator)
            top_v = acc[top_i]
        else:
            top_v, top_i = torch.topk(acc, K)
        parents = (top_i // V).tolist()
        toks = (top_i % V).tolist()
        last = step + 1 >= max_new_tokens
        ended = [bool(last or (eos and t in eos)) for t in toks]
        # running beams of the next step: the B best continuations that did not end
        masked = top_v + torch.tensor([NEG if e else 0.0 for e in ended], dtype=torch.float32, device=dev)
        nxt_v, nxt_j = torch.topk(masked, B)
        nxt_j = nxt_j.tolist()
        # finished: continuations among the top B that ended, normalised by their generated length
        cand = top_v / float((step + 1) ** length_penalty)
        full = all(fin_done) and early_stopping is True
        pen = [0.0 if (ended[j] and j < B and not full and improvable) else NEG for j in range(K)]
        cand = cand + torch.tensor(pen, dtype=torch.float32, device=dev)
        merged = torch.cat([fin_scores, cand])
        m_v, m_i = torch.topk(merged, B)
        new_fin_seqs, new_done = [], []
        for v, i in zip(m_v.tolist(), m_i.tolist()):
            if i < B:
                new_fin_seqs.append(fin_seqs[i]); new_done.append(fin_done[i])
            else:
                j = i - B
                new_fin_seqs.append(seqs[parents[j]] + [toks[j]]); new_done.append(ended[j] and j < B)
        fin_seqs, fin_scores, fin_done = new_fin_seqs, m_v, new_done
        new_seqs = [seqs[parents[j]] + [toks[j]] for j in nxt_j]
        next_parents, next_toks = [parents[j] for j in nxt_j], [toks[j] for j in nxt_j]
        seqs, run_scores = new_seqs, nxt_v
        # can a running beam still beat the worst finished one?  (early_stopping=False heuristic: best running score at the current length)
        cur = step + 1
        if early_stopping == "never" and length_penalty > 0.0:
            best_len = max_new_tokens
        else:
            best_len = cur
        best_running = float(run_scores[0]) / float(best_len ** length_penalty)
        worst_fin = float(fin_scores.min()) if any(fin_done) else NEG
        # (HF: worst_finished = where(is_sent_finished, min(beam_scores), -1e9) per finished slot; any(best > worst))
        improvable = improvable and any((best_running > (worst_fin if d else NEG)) for d in fin_done)
        exists_open = not (all(fin_done) and early_stopping is True)
        can_continue = not all(ended[:K]) if last else True
        if not (improvable and exists_open and not last and can_continue):
            break
        logits = advance(next_parents, next_toks)
    return fin_seqs[0]


class EngineBeams:
    """`advance` over a QuickPrefillEngine whose arena holds the prefilled video + prompt: beams share rows [0, P) of every layer; beam b owns
    a small tail block [L, 2, Hkv, T, D] with the K/V rows of its generated tokens."""

    def __init__(self, eng, rope_delta: int, num_beams: int, max_new_tokens: int):
        self.eng, self.delta, self.B = eng, rope_delta, num_beams
        self.base_len, self.base_pos = list(eng.arena.len), eng.seq_pos
        L, hkv, D = len(eng.arena.len), eng.hkv, eng.D
        self.tails = [torch.empty(L, 2, hkv, max_new_tokens, D, dtype=eng.dtype, device=eng.device) for _ in range(num_beams)]
        self.t = 0                                                     # generated tokens already cached per running beam

    def _load(self, tail):
        a, t = self.eng.arena, self.t
        for l, p in enumerate(self.base_len):
            if t:
                a.buf[l, :, :, p:p + t].copy_(tail[l, :, :, :t])
            a.len[l] = p + t
        self.eng.seq_pos = self.base_pos + t

    def advance(self, parents: List[int], tokens: List[int]) -> torch.Tensor:
        eng, t = self.eng, self.t
        new_tails = [torch.empty_like(self.tails[0]) for _ in range(self.B)]
        out = []
        for i, (par, tok) in enumerate(zip(parents, tokens)):
            self._load(self.tails[par])
            lg = eng.decode_step(eng.embed_tokens(torch.tensor([tok], device=eng.device)), self.delta)
            if lg is not None:                                          # (layer-pipeline stages other than the last hold no logits)
                out.append(lg.float())
            if t:
                new_tails[i][:, :, :, :t].copy_(self.tails[par][:, :, :, :t])
            for l, p in enumerate(self.base_len):                       # the row this step appended
                new_tails[i][l, :, :, t].copy_(eng.arena.buf[l, :, :, p + t])
        self.tails, self.t = new_tails, t + 1
        return torch.stack(out) if out else None

    def finish(self, generated: int):
        """Leave the engine with the winner's length bookkeeping (its rows beyond the shared prefix are not restored)."""
        for l, p in enumerate(self.base_len):
            self.eng.arena.len[l] = p
        self.eng.seq_pos = self.base_pos
