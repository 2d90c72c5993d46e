"""Tokenizer-side stand-in for the HF processor (no tokenizer files offline).

`SyntheticProcessor` reproduces what the engine needs from `processor.apply_chat_template` + tokenisation
(qwen25_lvu.py:546-548; the tokenizer-only `dummy_call` of the overlap plugin, interleaved:522-638): the Qwen2-VL
chat layout  <|im_start|>system ... <|im_start|>user <|vision_start|> <|video_pad|> x N <|vision_end|> question
<|im_end|> <|im_start|>assistant  with word-hash token ids.  A real HF tokenizer can be passed instead (anything
with encode/decode)."""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import List, Sequence

from .spec import TextSpec


def _fnv1a(s: str) -> int:
    h = 0xcbf29ce484222325
    for b in s.encode("utf-8"):
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


@dataclass
class Prompt:
    prefix_ids: List[int]     # everything before the first video token (incl. <|vision_start|>)
    tail_ids: List[int]       # <|vision_end|> + question + generation prompt


class SyntheticProcessor:
    def __init__(self, spec: TextSpec, tokenizer=None):
        self.spec, self.tokenizer = spec, tokenizer
        self.n_special = 64
        v = spec.vocab
        self.im_start, self.im_end = min(151644, v - 4), min(151645, v - 3)

    def encode(self, text: str) -> List[int]:
        if self.tokenizer is not None:
            return list(self.tokenizer.encode(text))
        words = re.findall(r"\w+|[^\w\s]", text)
        span = min(self.spec.vocab, 151000) - self.n_special - 16
        return [self.n_special + _fnv1a(w) % span for w in words]

    def decode(self, ids: Sequence[int]) -> str:
        if self.tokenizer is not None:
            return self.tokenizer.decode(list(ids))
        return " ".join(f"<tok_{int(i)}>" for i in ids)

    def batch_decode(self, batch, **kw) -> List[str]:
        return [self.decode(ids) for ids in batch]

    def build_prompt(self, question: str, system: str = "You are a helpful assistant.") -> Prompt:
        s = self.spec
        prefix = [self.im_start] + self.encode("system " + system) + [self.im_end, self.im_start] + self.encode("user") + [s.vision_start_token_id]
        tail = [s.vision_end_token_id] + self.encode(question) + [self.im_end, self.im_start] + self.encode("assistant")
        return Prompt(prefix, tail)
