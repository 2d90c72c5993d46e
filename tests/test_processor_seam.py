"""The processor seam (lvu/lvu.py:18-23; qwen25_lvu.py:546-548, 597-604): `LVU(config, model, processor)` with the kind of object a
reference user passes — an HF processor with `apply_chat_template` and a tokenizer.  No tokenizer files exist offline, so the tests
build a tiny `PreTrainedTokenizerFast` on the fly (byte-level BPE over printable ASCII, the Qwen2-VL special tokens as added tokens)
and put it, with the Qwen2-VL chat-template string, into the INSTALLED `transformers.Qwen2VLProcessor` (its video processor needs
torchvision, which the image lacks; a stub that only reports `video_grid_thw` stands in, so the processor's own
`<|video_pad|>`-expansion and tokenisation code is what the pipeline's ids are compared with)."""
import numpy as np
import pytest
import torch

from tests.oracle_ops import OracleOps

# the chat template shipped with the Qwen2-VL / Qwen2.5-VL checkpoints (chat_template.json [3P]); data, not code of the reference
QWEN2VL_TEMPLATE = (
    "{% set image_count = namespace(value=0) %}{% set video_count = namespace(value=0) %}{% for message in messages %}"
    "{% if loop.first and message['role'] != 'system' %}<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n{% endif %}"
    "<|im_start|>{{ message['role'] }}\n{% if message['content'] is string %}{{ message['content'] }}<|im_end|>\n{% else %}"
    "{% for content in message['content'] %}{% if content['type'] == 'image' or 'image' in content or 'image_url' in content %}"
    "{% set image_count.value = image_count.value + 1 %}{% if add_vision_id %}Picture {{ image_count.value }}: {% endif %}"
    "<|vision_start|><|image_pad|><|vision_end|>{% elif content['type'] == 'video' or 'video' in content %}"
    "{% set video_count.value = video_count.value + 1 %}{% if add_vision_id %}Video {{ video_count.value }}: {% endif %}"
    "<|vision_start|><|video_pad|><|vision_end|>{% elif 'text' in content %}{{ content['text'] }}{% endif %}{% endfor %}<|im_end|>\n"
    "{% endif %}{% endfor %}{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}")

# special ids of the tiny test models (TINY spec / the tiny HF checkpoints of test_api_cpu): video 300, vision_start 301, vision_end 302
SPECIALS = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<|image_pad|>", "<|video_pad|>", "<|vision_start|>", "<|vision_end|>"]   # ids 296..302


def tiny_hf_tokenizer():
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {chr(i + 33): i for i in range(94)}                 # printable ASCII as byte-level tokens
    vocab.update({"Ċ": 94, "Ġ": 95})                            # newline and space in the byte-level alphabet
    vocab.update({f"<f{i}>": 96 + i for i in range(200)})       # filler so that the added tokens start at id 296
    tk = Tokenizer(models.BPE(vocab=vocab, merges=[], unk_token=None))
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tk.decoder = decoders.ByteLevel()
    tk.add_special_tokens(SPECIALS)                             # ids 296..302 in this order
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="<|im_end|>", pad_token="<|endoftext|>", additional_special_tokens=SPECIALS)
    assert [tok.convert_tokens_to_ids(t) for t in SPECIALS] == list(range(296, 303))
    tok.chat_template = QWEN2VL_TEMPLATE
    return tok


def installed_qwen2vl_processor(grid_thw):
    """transformers' own Qwen2VLProcessor around the tiny tokenizer; `grid_thw` is what its (stubbed) video processor reports."""
    import transformers
    from transformers import BatchFeature, Qwen2VLImageProcessorPil, Qwen2VLProcessor

    class GridOnlyVideoProcessor(transformers.BaseVideoProcessor):      # the real one needs torchvision
        merge_size, temporal_patch_size = 2, 2

        def __init__(self):
            pass

        def __call__(self, videos=None, **kw):
            return BatchFeature({"video_grid_thw": torch.tensor([list(grid_thw)]), "pixel_values_videos": torch.zeros(1)})

    return Qwen2VLProcessor(image_processor=Qwen2VLImageProcessorPil(), tokenizer=tiny_hf_tokenizer(), video_processor=GridOnlyVideoProcessor(),
                            chat_template=QWEN2VL_TEMPLATE)


MESSAGES = [
    [{"role": "user", "content": [{"type": "video", "video": "v.mp4"}, {"type": "text", "text": "What happens in this video?"}]}],
    [{"role": "system", "content": "Answer in one word."},
     {"role": "user", "content": [{"type": "text", "text": "Look: "}, {"type": "video", "video": "v.mp4", "fps": 2}, {"type": "text", "text": "who wins?"}]}],
    [{"role": "user", "content": [{"type": "video", "video": "v.mp4"}, {"type": "text", "text": "Describe it."}]},
     {"role": "assistant", "content": "A match."},
     {"role": "user", "content": "Which teams?"}],
]


@pytest.mark.parametrize("mi", range(len(MESSAGES)))
def test_prompt_ids_equal_the_installed_processors(mi):
    """ids the pipeline builds (prefix | N video pads | tail) == `Qwen2VLProcessor(text=apply_chat_template(messages), videos=...)`
    input_ids: the installed processor expands `<|video_pad|>` to grid.prod() / merge^2 pads and tokenises the whole string
    (what the reference does, qwen25_lvu.py:597-604); multi-message conversations included (system turn, earlier assistant turn)."""
    from quickvideo_amd.processor import SyntheticProcessor, prompt_from_messages, qwen2vl_chat_text
    from quickvideo_amd.spec import TINY
    grid = (4, 8, 12)
    pr = installed_qwen2vl_processor(grid)
    msgs = MESSAGES[mi]
    text = pr.apply_chat_template(msgs, tokenize=False, add_generation_prompt=True)
    want = pr(text=[text], videos=[torch.zeros(8, 3, 28, 28)], return_tensors="pt")["input_ids"][0].tolist()
    p = prompt_from_messages(pr, msgs)
    n_video = grid[0] * grid[1] * grid[2] // 4
    assert p.prefix_ids + [300] * n_video + p.tail_ids == want
    assert p.prefix_ids[-1] == 301 and p.tail_ids[0] == 302                 # <|vision_start|> ... <|vision_end|>
    # the offline stand-in renders the SAME text (newlines included) and cuts at the same place
    assert qwen2vl_chat_text(msgs) == text
    sp = SyntheticProcessor(TINY)
    q = prompt_from_messages(sp, msgs)
    assert q.prefix_ids[-1] == 301 and q.tail_ids[0] == 302 and sp.newline_id in q.prefix_ids and q.tail_ids[-1] == sp.newline_id
    # plugging the real tokenizer INTO the stand-in gives the reference ids too
    sp2 = SyntheticProcessor(TINY, tokenizer=type("T", (), {"encode": staticmethod(lambda t: pr.tokenizer(t)["input_ids"]),
                                                            "decode": staticmethod(lambda ids: pr.tokenizer.decode(ids))})())
    sp2.special = {t: pr.tokenizer.convert_tokens_to_ids(t) for t in sp2.special}
    r = prompt_from_messages(sp2, msgs)
    assert (r.prefix_ids, r.tail_ids) == (p.prefix_ids, p.tail_ids)


def test_processor_without_template_is_refused():
    from quickvideo_amd.processor import prompt_from_messages
    with pytest.raises(TypeError):
        prompt_from_messages(object(), MESSAGES[0])
    pr = installed_qwen2vl_processor((2, 4, 4))
    two = [{"role": "user", "content": [{"type": "video", "video": "a"}, {"type": "video", "video": "b"}, {"type": "text", "text": "?"}]}]
    with pytest.raises(AssertionError, match="Only one video"):
        prompt_from_messages(pr, two)


def test_lvu_chat_with_hf_processor_object(capsys, monkeypatch):
    """`LVU(config, model, processor)` with the HF processor object, multi-message `chat()`: the engine is fed exactly the ids the
    processor's template + tokenizer produce, answers come back through the processor's `batch_decode`, generation stops at the
    tokenizer's EOS, and the repetition penalty sees the TAIL ids only — the reference calls HF generate with
    `input_ids[:, past_len:]` over the pre-filled cache (qwen25_lvu.py:724-740), so HF's processors never see the system prompt or
    the video pads."""
    import lvu
    from quickvideo_amd.lvu import load_native_model
    from quickvideo_amd.pipeline import PrefillPipeline
    from quickvideo_amd.processor import prompt_from_messages
    from quickvideo_amd.sampling import TokenSelector
    m = load_native_model("synthetic:tiny", device="cpu")
    pr = installed_qwen2vl_processor((4, 8, 12))
    obj = lvu.LVU(lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=4, num_frames=8), model=m, processor=pr)
    obj._ops = OracleOps()
    video = "synthetic://?frames=32&h=112&w=168&seed=2&pattern=gradient"
    msgs = [{"role": "system", "content": "Answer briefly."},
            {"role": "user", "content": [{"type": "video", "video": video}, {"type": "text", "text": "What is shown?"}]}]
    seen_prompts, observed = [], []
    orig_plan, orig_obs = PrefillPipeline.plan, TokenSelector.observe

    def plan(self, reader, question):
        P = orig_plan(self, reader, question)
        seen_prompts.append(P["prompt"])
        return P

    def observe(self, ids, vocab, device):
        observed.append(list(ids))
        return orig_obs(self, ids, vocab, device)
    monkeypatch.setattr(PrefillPipeline, "plan", plan)
    monkeypatch.setattr(TokenSelector, "observe", observe)
    out = obj.chat(msgs, max_new_tokens=4, eos_token_id=None, repetition_penalty=1.3)
    assert isinstance(out, list) and len(out) == 1 and isinstance(out[0], str)
    want = prompt_from_messages(pr, msgs)
    assert (seen_prompts[0].prefix_ids, seen_prompts[0].tail_ids) == (want.prefix_ids, want.tail_ids)
    assert observed and observed[0] == want.tail_ids                      # penalty history = tail only (then one id per generated token)
    assert all(len(o) == 1 for o in observed[1:])
    # default EOS = the tokenizer's (<|im_end|> = 298): force it as the first token -> one token, decoded to "" (special skipped)
    monkeypatch.setattr(TokenSelector, "select", lambda self, logits: 298)
    monkeypatch.setattr(TokenSelector, "trivial", property(lambda self: False))
    out2 = obj.chat(msgs, max_new_tokens=4)
    assert out2 == [""]
    # generate(question, video) builds the one-video message itself (qwen25_lvu.py:504-536)
    monkeypatch.undo()
    out3 = obj.generate("What is shown?", video, max_new_tokens=2, eos_token_id=None)
    assert len(out3) == 1


def test_eos_list_from_generation_config_stops_generation():
    """HF generate stops on ANY id of generation_config.eos_token_id (Qwen2/2.5-VL ship [151645, 151643]); a list must work in both
    decode loops (GraphDecoder.generate takes the same set on the GPU)."""
    import lvu
    from quickvideo_amd.lvu import load_native_model
    m = load_native_model("synthetic:tiny", device="cpu")
    obj = lvu.LVU(lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=4, num_frames=8), model=m)
    obj._ops = OracleOps()
    video = "synthetic://?frames=32&h=112&w=168&seed=2&pattern=gradient"
    from quickvideo_amd.pipeline import PrefillPipeline
    pipe = PrefillPipeline(m, obj.config, obj.processor, ops=OracleOps())
    ids = pipe.generate("What?", video, max_new_tokens=4, eos_token_id=None)
    assert len(ids) == 4
    j = next((i for i in range(4) if ids[i] not in ids[:i] and i > 0), None)          # first position holding a NEW id
    if j is not None:
        assert pipe.generate("What?", video, max_new_tokens=4, eos_token_id=[9999, ids[j]]) == ids[:j + 1]
    assert pipe.generate("What?", video, max_new_tokens=4, eos_token_id=(9999, ids[0])) == ids[:1]
    m.generation_defaults = {"eos_token_id": [ids[0], 7]}                  # from generation_config.json
    assert pipe.generate("What?", video, max_new_tokens=4) == ids[:1]
    m.generation_defaults = None


@pytest.mark.parametrize("fps,total,nframes", [(24.0, 28416, 768), (30.0, 5400, 360), (25.0, 1000, 64), (29.97, 2997, 200), (2.0, 64, 64), (3.0, 90, 20)])
def test_qwen25_temporal_ids_equal_hf_float32(fps, total, nframes):
    """Qwen2.5-VL temporal M-RoPE ids as transformers==4.50.0 (the reference's pin, uv.lock:1380-1381) computes them in
    get_rope_index [3P]:  `range_tensor.expand(-1, h*w) * second_per_grid_t * tokens_per_second` -> `.long()`, with second_per_grid_t
    an element of the FLOAT32 tensor `second_per_grid_ts` — so both products are rounded to fp32 before the truncation.  The
    planner mirrors that; the same expression in float64 differs by one on ~2 % of (length, nframes) pairs (the first case).  Checked
    against the expression evaluated by torch itself.  (The installed transformers 5.15 changed the rule — it truncates
    second_per_grid_t to an integer interval first — so its get_rope_index is not the oracle for this detail.)"""
    from quickvideo_amd import planner
    sample_fps = nframes / max(total, 1e-6) * fps                    # qwen-vl-utils: video_sample_fps
    t, gh, gw, prefix, tail = nframes // 2, 4, 4, 5, 3
    hw = (gh // 2) * (gw // 2)
    spg = torch.tensor([2 / sample_fps])[0]                           # float32, like BatchFeature's tensor conversion of the python float
    want_t = (torch.arange(t).view(-1, 1).expand(-1, hw) * spg * 2).long().flatten().numpy()
    got, d = planner.mrope_positions(prefix, (t, gh, gw), tail, second_per_grid_t=2 / sample_fps, tokens_per_second=2)
    assert np.array_equal(got[0, prefix:prefix + t * hw] - prefix, want_t)
    f64 = (np.arange(t, dtype=np.float64) * (2 * 2 / sample_fps)).astype(np.int64)
    if (fps, total, nframes) == (24.0, 28416, 768):
        assert not np.array_equal(np.repeat(f64, hw), want_t), "the float64 evaluation is the one that differs on this case"
