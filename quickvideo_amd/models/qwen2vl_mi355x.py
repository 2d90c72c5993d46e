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


def _the_video(messages):
    """The single video entry of the conversation (qwen25_lvu.py:552-554: `extract_vision_info`, one video only)."""
    videos = [c["video"] for m in messages if not isinstance(m["content"], str) for c in m["content"]
              if c.get("type") == "video" or "video" in c]
    assert len(videos) == 1, "Only one video is supported for now."
    return videos[0]


def chat_lvu_model(self, messages, _overlap: bool = True, **generation_kwargs):
    """`messages` goes to the processor's chat template as it is (system / user / assistant turns, several text entries): the
    reference renders `processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)` (qwen25_lvu.py:546-548).
    `_overlap` is how the sequential plugin (qwen2vl_mi355x_sequential) reuses this function: an argument, not a module global,
    so two LVU objects of different plugins can generate concurrently."""
    video = _the_video(messages)
    pipe = getattr(self, "_pipeline", None)
    if pipe is None or pipe.cfg is not self.config or pipe.model is not self.model or pipe.processor is not self.processor:
        pipe = PrefillPipeline(self.model, self.config, self.processor, ops=getattr(self, "_ops", None))
        self._pipeline = pipe
    mnt = generation_kwargs.pop("max_new_tokens", 16)
    if "eos_token_id" not in generation_kwargs:
        # HF generate stops on ANY id of the checkpoint's generation_config.eos_token_id ([<|im_end|>, <|endoftext|>] for
        # Qwen2/2.5-VL; qwen25_lvu.py:740); without one: the tokenizer's EOS, else <|im_end|> (a chat turn ends with it)
        eos = (getattr(self.model, "generation_defaults", None) or {}).get("eos_token_id")
        if eos is None:
            eos = getattr(getattr(self.processor, "tokenizer", None), "eos_token_id", None)
        if eos is None:
            eos = getattr(self.processor, "eos_token_id", None)
        if eos is None:
            eos = getattr(self.processor, "im_end", None)
        generation_kwargs["eos_token_id"] = eos
    ids = pipe.generate(messages, video, max_new_tokens=mnt, overlap=_overlap, **generation_kwargs)
    t = pipe.last_timings
    if pipe.par.on and pipe.par.rank != 0:               # multi-GPU job: every rank returns the answer, rank 0 reports the timings
        return self.processor.batch_decode([ids], skip_special_tokens=True, clean_up_tokenization_spaces=False)
    # the reference prints these six lines (qwen25_lvu.py:748-753) from unsynchronised host clocks; here: the producer's time inside
    # the frame source, the ViT by itself, the device-synchronised group loop, decode, e2e, first token
    print(f"total time spent fetching frames was: {t.sequential_fetch if not _overlap else t.producer_busy}")
    print(f"total time spent on processor was: {t.vit_uncontended or t.vit_span}")     # GPU patchify + ViT (its span on the ViT stream)
    print(f"total time spent on prefill was: {t.prefill}")
    print(f"total time spent on decoding was: {t.decode}")
    print(f"total time spent on e2e fetching and decoding was: {t.e2e}")
    print(f"time to first token was: {t.ttft}")
    return self.processor.batch_decode([ids], skip_special_tokens=True, clean_up_tokenization_spaces=False)


def video_message(config: LVUConfig, question, video_path):
    """The one-video chat message run_lvu_model builds (qwen25_lvu.py:504-536): the video entry carries max_pixels / min_pixels from
    `extra_kwargs` and `fps` (if set) XOR `nframes` from the config — chat_lvu_model then reads ONLY the entry, like the reference."""
    from ..planner import video_entry_from_config
    cfg = config
    entry = video_entry_from_config(video_path, cfg.fps, cfg.num_frames, cfg.extra_kwargs)      # raises "Either fps or num_frames should be set."
    return [{"role": "user", "content": [entry, {"type": "text", "text": question}]}]


def run_lvu_model(self, question, video_path, **generation_kwargs):
    return chat_lvu_model(self, video_message(self.config, question, video_path), **generation_kwargs)
