"""Next-token selection for the decode leg (a10): what HF `generate` does to the logits of every step in the reference
(`qwen25_lvu.py:744-761` passes `**generation_kwargs` straight to `model.generate`; transformers `generation/logits_process.py` [3P]).
Qwen2.5-VL's shipped generation config is `do_sample=True, temperature=1e-6, top_k=1, top_p=0.001, repetition_penalty=1.05`, i.e.
greedy after a repetition penalty — so the penalty matters even for "greedy" users of the reference.

Order as in HF: processors (repetition penalty) -> [sampling only] warpers (temperature, top-k, top-p) -> softmax -> multinomial;
without `do_sample` the argmax of the processed scores.  Pure torch on the 152k-entry logits vector of one step (device-agnostic;
pinned against the installed transformers classes in tests/test_api_cpu.py).  Beam search: quickvideo_amd/beam.py.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch


class TokenSelector:
    def __init__(self, do_sample: bool = False, temperature: Optional[float] = None, top_k: Optional[int] = None, top_p: Optional[float] = None,
                 repetition_penalty: Optional[float] = None, seed: Optional[int] = None, device="cpu"):
        self.do_sample = bool(do_sample)
        self.temperature = 1.0 if temperature is None else float(temperature)
        self.top_k = 0 if top_k is None else int(top_k)
        self.top_p = 1.0 if top_p is None else float(top_p)
        self.penalty = 1.0 if repetition_penalty is None else float(repetition_penalty)
        if self.do_sample and not self.temperature > 0:
            raise ValueError(f"`temperature` (={temperature}) has to be a strictly positive float")          # HF's own message
        if not 0 <= self.top_p <= 1.0:
            raise ValueError(f"`top_p` has to be a float > 0 and < 1, but is {top_p}")
        if self.top_k < 0:
            raise ValueError(f"`top_k` has to be a strictly positive integer, but is {top_k}")
        if not self.penalty > 0:
            raise ValueError(f"`penalty` has to be a strictly positive float, but is {repetition_penalty}")
        self.gen = None
        if self.do_sample:
            self.gen = torch.Generator(device=device)
            if seed is not None:
                self.gen.manual_seed(int(seed))
            else:
                self.gen.seed()
        self.seen: Optional[torch.Tensor] = None          # bool [V]: ids that occurred in the prompt or were generated

    @property
    def trivial(self) -> bool:
        """True when the choice is a plain argmax of the raw logits (the hipGraph decoder keeps that on the device)."""
        return self.penalty == 1.0 and (not self.do_sample or self.top_k == 1)

    def observe(self, ids: Iterable[int], vocab: int, device):
        if self.penalty == 1.0:
            return
        if self.seen is None:
            self.seen = torch.zeros(vocab, dtype=torch.bool, device=device)
        idx = torch.as_tensor(list(ids), dtype=torch.long, device=device)
        if idx.numel():
            self.seen[idx[idx < vocab]] = True

    def process(self, logits: torch.Tensor) -> torch.Tensor:
        """fp32 scores after processors and (when sampling) warpers; filtered entries are -inf."""
        s = logits.float().clone()
        if self.penalty != 1.0 and self.seen is not None:                 # RepetitionPenaltyLogitsProcessor
            pen = torch.where(s < 0, s * self.penalty, s / self.penalty)
            s = torch.where(self.seen, pen, s)
        if not self.do_sample:
            return s
        if self.temperature != 1.0:                                       # TemperatureLogitsWarper
            s = s / self.temperature
        if self.top_k > 0:                                                # TopKLogitsWarper
            k = min(self.top_k, s.numel())
            kth = torch.topk(s, k).values[-1]
            s = s.masked_fill(s < kth, float("-inf"))
        if self.top_p < 1.0:                                              # TopPLogitsWarper (ascending sort, keep >= 1 token)
            sorted_s, order = torch.sort(s, descending=False)
            cum = sorted_s.softmax(-1).cumsum(-1)
            remove = cum <= (1.0 - self.top_p)
            remove[-1:] = False
            s = s.masked_fill(torch.zeros_like(remove).scatter(0, order, remove), float("-inf"))
        return s

    def select(self, logits: torch.Tensor) -> int:
        s = self.process(logits)
        if not self.do_sample:
            tok = int(torch.argmax(s).item())
        else:
            tok = int(torch.multinomial(torch.softmax(s, -1), 1, generator=self.gen).item())
        if self.seen is not None:
            self.seen[tok] = True
        return tok
