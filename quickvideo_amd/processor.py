"""The processor seam: `LVU(config, model, processor)` takes an HF `AutoProcessor` in the reference (lvu/lvu.py:18-23) and builds
its prompt with `processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)` followed by the
tokenizer, every `<|video_pad|>` expanded to one pad per merged vision token (qwen25_lvu.py:546-548, 597-604; tokenizer-only
twin `dummy_call`, qwen25_lvu_interleaved.py:522-638).

`prompt_from_messages` does exactly that with ANY object that has `apply_chat_template` and a tokenizer — the HF processor a
reference user already holds, or the offline stand-in below — and returns the ids in front of and behind the video pads.  The
pads themselves never need to exist as ids: their count follows from (nframes, H, W) (planner.py) and the engine takes the
vision tower's rows for them.

`SyntheticProcessor` is the offline fallback (no tokenizer files in this image): the Qwen2-VL chat template restated in Python
(same layout, newline tokens included) over a word-hash tokenizer that treats the template's special tokens like a HF tokenizer
treats added tokens.  A real tokenizer can be plugged into it (`tokenizer=`), or a whole HF processor passed to `LVU` instead."""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import List, Sequence

from .spec import TextSpec

VIDEO_TOKEN, IMAGE_TOKEN = "<|video_pad|>", "<|image_pad|>"


def _fnv1a(s: str) -> int:
    h = 0xcbf29ce484222325
    for b in s.encode("utf-8"):
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


@dataclass
class Prompt:
    prefix_ids: List[int]     # everything before the first video token (incl. <|vision_start|>)
    tail_ids: List[int]       # <|vision_end|> + question + generation prompt


def as_messages(question_or_messages, video="<video>"):
    """A bare question -> the one-video user message `run_lvu_model` builds (qwen25_lvu.py:504-536)."""
    if isinstance(question_or_messages, str):
        return [{"role": "user", "content": [{"type": "video", "video": video}, {"type": "text", "text": question_or_messages}]}]
    return question_or_messages


def qwen2vl_chat_text(messages, add_generation_prompt: bool = True, add_vision_id: bool = False) -> str:
    """The Qwen2-VL / Qwen2.5-VL chat template (the `chat_template` shipped with those checkpoints [3P]) as plain Python: default
    system turn unless the first message is a system message; per message `<|im_start|>{role}\\n ... <|im_end|>\\n`; an image /
    video entry becomes `<|vision_start|><|image_pad|>|<|video_pad|><|vision_end|>`; text entries are pasted as they are."""
    out, n_img, n_vid = [], 0, 0
    for i, m in enumerate(messages):
        if i == 0 and m["role"] != "system":
            out.append("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n")
        out.append(f"<|im_start|>{m['role']}\n")
        if isinstance(m["content"], str):
            out.append(m["content"])
        else:
            for c in m["content"]:
                if c.get("type") == "image" or "image" in c or "image_url" in c:
                    n_img += 1
                    out.append((f"Picture {n_img}: " if add_vision_id else "") + "<|vision_start|>" + IMAGE_TOKEN + "<|vision_end|>")
                elif c.get("type") == "video" or "video" in c:
                    n_vid += 1
                    out.append((f"Video {n_vid}: " if add_vision_id else "") + "<|vision_start|>" + VIDEO_TOKEN + "<|vision_end|>")
                elif "text" in c:
                    out.append(c["text"])
        out.append("<|im_end|>\n")
    if add_generation_prompt:
        out.append("<|im_start|>assistant\n")
    return "".join(out)


def _tokenizer_of(processor):
    tok = getattr(processor, "tokenizer", None)
    return tok if tok is not None else processor


def _encode(tok, text: str) -> List[int]:
    """ids of `text` exactly as the reference's `self.tokenizer(text)` yields them (dummy_call, interleaved:636)."""
    if callable(tok) and not isinstance(tok, SyntheticProcessor):
        enc = tok(text)
        ids = enc["input_ids"] if not isinstance(enc, list) else enc
        return [int(i) for i in (ids[0] if ids and isinstance(ids[0], (list, tuple)) else ids)]
    return [int(i) for i in tok.encode(text)]


def prompt_from_messages(processor, messages) -> Prompt:
    """messages -> (ids before the video pads, ids behind them), through the processor's own chat template and tokenizer.

    The template is rendered with ONE `<|video_pad|>` per video entry, like `apply_chat_template` leaves it; the reference then
    replaces it by N pads and tokenises (qwen25_lvu.py:597-604).  The pad is an added (special) token, so the tokens on either
    side of it do not depend on N: tokenising the rendered text once and cutting at the pad gives the reference's ids."""
    if not hasattr(processor, "apply_chat_template"):
        raise TypeError("processor needs apply_chat_template(messages, tokenize=False, add_generation_prompt=True) and a tokenizer "
                        "(an HF AutoProcessor, or quickvideo_amd.processor.SyntheticProcessor)")
    text = processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
    if isinstance(text, (list, tuple)):
        assert len(text) == 1, "Only batch size 1 is supported (utils.py:264)"
        text = text[0]
    vtok = getattr(processor, "video_token", VIDEO_TOKEN)
    assert text.count(vtok) == 1, "Only one video is supported for now."                    # qwen25_lvu.py:554
    tok = _tokenizer_of(processor)
    ids = _encode(tok, text)
    if hasattr(tok, "convert_tokens_to_ids"):
        vid = int(tok.convert_tokens_to_ids(vtok))
    else:
        vid = int(processor.video_token_id)
    at = [i for i, t in enumerate(ids) if t == vid]
    if len(at) != 1:
        raise ValueError(f"the tokenizer does not map {vtok!r} to one id of its own ({len(at)} hits): the video placeholder cannot be located")
    return Prompt(ids[:at[0]], ids[at[0] + 1:])


class SyntheticProcessor:
    """Offline processor + tokenizer: Qwen2-VL chat layout, word-hash ids (the template's special tokens and the newline get the
    ids the Qwen2 tokenizer gives them when the vocabulary is large enough)."""
    video_token, image_token = VIDEO_TOKEN, IMAGE_TOKEN

    def __init__(self, spec: TextSpec, tokenizer=None):
        self.spec, self._ext = spec, tokenizer
        self.n_special = 64
        v = spec.vocab
        big = v > 151700
        self.im_start, self.im_end = (151644, 151645) if big else (v - 4, v - 3)
        self.newline_id = 198 if big else self.n_special - 1
        self.special = {"<|im_start|>": self.im_start, "<|im_end|>": self.im_end, "<|vision_start|>": spec.vision_start_token_id,
                        "<|vision_end|>": spec.vision_end_token_id, VIDEO_TOKEN: spec.video_token_id,
                        IMAGE_TOKEN: getattr(spec, "image_token_id", spec.video_token_id - 1)}
        self._split = re.compile("(" + "|".join(re.escape(t) for t in self.special) + ")")
        self.video_token_id = spec.video_token_id
        self.eos_token_id = self.im_end
        self.tokenizer = self                     # the seam's shape: processor.tokenizer(text) / .encode / .decode

    # -- tokenizer side
    def _words(self, text: str) -> List[int]:
        if self._ext is not None:
            return list(self._ext.encode(text))
        span = min(self.spec.vocab, 151000, min(self.special.values())) - self.n_special      # word ids never collide with a special id
        out = []
        for w in re.findall(r"\n|\w+|[^\w\s]", text):
            out.append(self.newline_id if w == "\n" else self.n_special + _fnv1a(w) % span)
        return out

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for piece in self._split.split(text):
            if piece in self.special:
                ids.append(self.special[piece])
            elif piece:
                ids.extend(self._words(piece))
        return ids

    def __call__(self, text, **kw):
        return {"input_ids": self.encode(text if isinstance(text, str) else text[0])}

    def convert_tokens_to_ids(self, token: str) -> int:
        return self.special[token]

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = False, **kw) -> str:
        sp = set(self.special.values())
        ids = [int(i) for i in ids if not (skip_special_tokens and int(i) in sp)]
        if self._ext is not None:
            return self._ext.decode(ids)
        return " ".join(f"<tok_{i}>" for i in ids)

    def batch_decode(self, batch, **kw) -> List[str]:
        return [self.decode(ids, **kw) for ids in batch]

    # -- processor side
    def apply_chat_template(self, messages, tokenize: bool = False, add_generation_prompt: bool = True, **kw):
        text = qwen2vl_chat_text(messages, add_generation_prompt=add_generation_prompt)
        return self.encode(text) if tokenize else text

    def build_prompt(self, question: str, system: str = None) -> Prompt:
        """The one-video prompt around `question` (kept for callers of rounds 1-2; same result as the message route)."""
        msgs = as_messages(question)
        if system is not None:
            msgs = [{"role": "system", "content": system}] + msgs
        return prompt_from_messages(self, msgs)
