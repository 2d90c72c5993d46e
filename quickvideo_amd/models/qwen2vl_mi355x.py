"""Native MI355X plugin (overlapped CPU frame production + GPU ViT/prefill) — the counterpart of the reference's
lvu/models/qwen25_lvu_interleaved.py behind the same init/run/chat triple."""
from ..lvu_config import LVUConfig
from ..pipeline import PrefillPipeline, QwenVLNative


def init_lvu_model(model: QwenVLNative, config: LVUConfig):
    """Reference: rebinding every decoder layer's forward (qwen25_lvu.py:467-502).  Here the model object already is
    the native weight set; this attaches the LVU configuration the engine reads."""
    if not isinstance(model, QwenVLNative):
        raise ValueError("the native plugin expects a quickvideo_amd.pipeline.QwenVLNative model "
                         "(see quickvideo_amd.lvu.load_native_model)")
    model.config = config
    model.engine = None
    return model


def _content_question(messages):
    video, question = None, ""
    for m in messages:
        for c in (m["content"] if isinstance(m["content"], list) else [{"type": "text", "text": m["content"]}]):
            if c.get("type") == "video":
                assert video is None, "Only one video is supported for now."      # qwen25_lvu.py:554
                video = c["video"]
            elif c.get("type") == "text":
                question += c["text"]
    assert video is not None, "Only one video is supported for now."
    return video, question


def chat_lvu_model(self, messages, _overlap: bool = True, **generation_kwargs):
    """`_overlap` is how the sequential plugin (qwen2vl_mi355x_sequential) reuses this function: an argument, not a module global,
    so two LVU objects of different plugins can generate concurrently."""
    video, question = _content_question(messages)
    pipe = getattr(self, "_pipeline", None)
    if pipe is None or pipe.cfg is not self.config or pipe.model is not self.model:
        pipe = PrefillPipeline(self.model, self.config, self.processor, ops=getattr(self, "_ops", None))
        self._pipeline = pipe
    mnt = generation_kwargs.pop("max_new_tokens", 16)
    if "eos_token_id" not in generation_kwargs:              # HF generate stops at the generation config's EOS (qwen25_lvu.py:740)
        eos = getattr(self.processor, "eos_token_id", None)
        if eos is None:
            eos = getattr(getattr(self.processor, "tokenizer", None), "eos_token_id", None)
        if eos is None:
            eos = getattr(self.processor, "im_end", None)   # Qwen2-VL chat models end a turn with <|im_end|>
        generation_kwargs["eos_token_id"] = eos
    ids = pipe.generate(question, video, max_new_tokens=mnt, overlap=_overlap, **generation_kwargs)
    t = pipe.last_timings
    # the reference prints these six lines (qwen25_lvu.py:748-753); ours are device-synchronised
    print(f"total time spent fetching frames was: {t.fetch}")
    print(f"total time spent on processor was: {t.vit}")
    print(f"total time spent on prefill was: {t.prefill}")
    print(f"total time spent on decoding was: {t.decode}")
    print(f"total time spent on e2e fetching and decoding was: {t.e2e}")
    print(f"time to first token was: {t.ttft}")
    return self.processor.batch_decode([ids], skip_special_tokens=True, clean_up_tokenization_spaces=False)


def run_lvu_model(self, question, video_path, **generation_kwargs):
    """qwen25_lvu.py:504-536: one-video chat message; `fps` xor `num_frames` comes from the config."""
    cfg = self.config
    if cfg.fps is None and cfg.num_frames is None:
        raise ValueError("Either fps or num_frames should be set.")
    messages = [{"role": "user", "content": [{"type": "video", "video": video_path}, {"type": "text", "text": question}]}]
    return chat_lvu_model(self, messages, **generation_kwargs)
