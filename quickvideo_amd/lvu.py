"""`LVU` — the reference's public class (lvu/lvu.py:7-57) kept drop-in: LVU(config, model=None, processor=None,
model_init_kwargs={}), .generate(question, video_path, **gen) -> list[str], .chat(messages, **gen) -> list[str]."""
import os

import torch

from .lvu_config import LVUConfig
from .models import lvu_chat_model_map, lvu_init_model_map, lvu_run_model_map
from .pipeline import QwenVLNative
from .processor import SyntheticProcessor
from .spec import PRESETS
from .vit import QWEN25_VL_VIT_7B, QWEN2_VL_VIT_2B, QWEN2_VL_VIT_72B, QWEN2_VL_VIT_7B, TINY_VIT, VisionWeights
from .weights import DecoderWeights

_VIT = {"qwen2-vl-2b": QWEN2_VL_VIT_2B, "qwen2-vl-7b": QWEN2_VL_VIT_7B, "qwen2.5-vl-7b": QWEN25_VL_VIT_7B, "qwen2-vl-72b": QWEN2_VL_VIT_72B,
        "tiny": TINY_VIT}


def load_native_model(name_or_path: str, device=None, seed: int = 0, parallel=None, process_group=None) -> QwenVLNative:
    """`synthetic:<preset>` -> seeded random weights at the real dims (no checkpoints offline), or a local
    directory with a HF Qwen2-VL checkpoint (config.json + *.safetensors).

    Multi-GPU (one process per GPU, torch.distributed initialised by the launcher; the reference's counterpart is
    `device_map="auto"`, lvu/lvu.py:11-16): `parallel` = "tp" | "sp" | "pp" | "auto" | "single" (default: $QP_PARALLEL, else "tp" in a
    multi-rank job).  "tp" loads this rank's head / MLP-column SHARD; the other modes load a full replica on every rank and cut the
    per-video pipeline stages out of it as views (quickvideo_amd/parallel.py).  The vision tower is replicated in every mode."""
    from .parallel import resolve
    par = resolve(parallel, process_group)
    if device is None:
        device = (f"cuda:{torch.cuda.current_device()}" if par.on else "cuda:0") if torch.cuda.is_available() else "cpu"
    device = torch.device(device)
    tp = dict(tp_rank=par.rank, tp_size=par.world) if (par.on and par.mode == "tp") else {}
    key = name_or_path.split(":", 1)[1] if name_or_path.startswith("synthetic:") else None
    if key is None:
        low = os.path.basename(name_or_path.rstrip("/")).lower()
        if os.path.isdir(name_or_path):
            m = _load_hf_dir(name_or_path, device, **tp)
            m.parallel = par
            return m
        key = next((k for k in PRESETS if k.replace("-", "") in low.replace("-", "").replace("instruct", "")), None)
        if key is None:
            raise ValueError(f"cannot resolve model {name_or_path!r}: use 'synthetic:<{'|'.join(PRESETS)}>' or a local checkpoint directory")
    spec = PRESETS[key]
    return QwenVLNative(DecoderWeights.synthetic(spec, device, seed=seed, **tp), VisionWeights.synthetic(_VIT[key], device, seed=seed), device, name=key,
                        parallel=par)


def _load_hf_dir(path: str, device, tp_rank: int = 0, tp_size: int = 1) -> QwenVLNative:
    import json
    from safetensors import safe_open
    from .spec import TextSpec
    from .vit import VisionSpec
    cfg = json.load(open(os.path.join(path, "config.json")))
    tc = cfg.get("text_config", cfg)
    spec = TextSpec(hidden=tc["hidden_size"], n_heads=tc["num_attention_heads"], n_kv_heads=tc["num_key_value_heads"],
                    head_dim=tc["hidden_size"] // tc["num_attention_heads"], intermediate=tc["intermediate_size"],
                    n_layers=tc["num_hidden_layers"], vocab=tc["vocab_size"], rope_theta=tc.get("rope_theta", 1e6),
                    mrope_section=tuple((tc.get("rope_scaling") or tc.get("rope_parameters") or {}).get("mrope_section", (16, 24, 24))),
                    rms_eps=tc.get("rms_norm_eps", 1e-6), tie_embeddings=cfg.get("tie_word_embeddings", False),
                    video_token_id=cfg.get("video_token_id", 151656), vision_start_token_id=cfg.get("vision_start_token_id", 151652),
                    vision_end_token_id=cfg.get("vision_end_token_id", 151653))
    vc = cfg["vision_config"]
    if "embed_dim" in vc:                                     # Qwen2-VL tower
        vspec = VisionSpec(depth=vc["depth"], embed_dim=vc["embed_dim"], num_heads=vc["num_heads"], mlp_ratio=vc.get("mlp_ratio", 4),
                           patch_size=vc.get("patch_size", 14), temporal_patch_size=vc.get("temporal_patch_size", 2),
                           spatial_merge_size=vc.get("spatial_merge_size", 2), out_hidden=vc.get("hidden_size", spec.hidden))
    else:                                                     # Qwen2.5-VL tower (the reference's family, lvu.py:60): hidden_size is the ViT width
        vspec = VisionSpec(arch="qwen2.5", depth=vc["depth"], embed_dim=vc["hidden_size"], num_heads=vc["num_heads"],
                           patch_size=vc.get("patch_size", 14), temporal_patch_size=vc.get("temporal_patch_size", 2),
                           spatial_merge_size=vc.get("spatial_merge_size", 2), out_hidden=vc.get("out_hidden_size", spec.hidden),
                           intermediate=vc["intermediate_size"], window_size=vc.get("window_size", 112),
                           fullatt_blocks=tuple(vc.get("fullatt_block_indexes", (7, 15, 23, 31))))
        # temporal M-RoPE ids advance by second_per_grid_t * tokens_per_second per temporal patch (get_rope_index [3P]); the
        # pipeline derives second_per_grid_t from the sampled fps; -1 = "per video" marker resolved in PrefillPipeline.plan
        from dataclasses import replace
        spec = replace(spec, temporal_scale=-float(vc.get("tokens_per_second", 2)))
    sd = {}
    for f in sorted(os.listdir(path)):
        if f.endswith(".safetensors"):
            with safe_open(os.path.join(path, f), "pt") as sf:
                for k in sf.keys():
                    sd[k] = sf.get_tensor(k)
    text = {k.split("model.", 1)[-1].replace("language_model.", ""): v for k, v in sd.items() if "visual" not in k}
    vis = {k.split("visual.", 1)[1]: v for k, v in sd.items() if "visual." in k}
    gen = None
    gpath = os.path.join(path, "generation_config.json")           # HF `generate` applies it implicitly (Qwen2.5-VL ships do_sample /
    if os.path.exists(gpath):                                       # temperature 1e-6 / top_k 1 / top_p 0.001 / repetition_penalty 1.05)
        g = json.load(open(gpath))
        gen = {k: g[k] for k in ("do_sample", "temperature", "top_k", "top_p", "repetition_penalty", "eos_token_id") if k in g}
    return QwenVLNative(DecoderWeights.from_named(spec, text, device, tp_rank=tp_rank, tp_size=tp_size), VisionWeights.from_named(vspec, vis, device),
                        device, name=path, generation_defaults=gen)


class LVU:
    def __init__(self, config, model=None, processor=None, model_init_kwargs={}):
        self.config = config
        if model is None:
            # reference: AutoModelForImageTextToText.from_pretrained(bf16, device_map="auto", flash_attention_2) (lvu.py:10-16)
            # plus, for a multi-GPU job (one process per GPU): parallel="tp"|"sp"|"pp"|"auto", process_group=<group of the job's ranks>
            model = load_native_model(config.model_name_or_path, **{k: v for k, v in model_init_kwargs.items()
                                                                    if k in ("device", "seed", "parallel", "process_group")})
        if processor is None:
            processor = SyntheticProcessor(model.spec)           # reference: AutoProcessor.from_pretrained (lvu.py:19-20)
        self.model = model
        self.processor = processor
        self.model = self.init_lvu()

    def run_model_func(self, question, video_path, **generation_kwargs):
        raise NotImplementedError("run_model_func not implemented.")

    def chat_model_func(self, messages, **generation_kwargs):
        raise NotImplementedError("chat_model_func not implemented.")

    def init_lvu(self):
        if self.config.model_type not in lvu_init_model_map:
            raise ValueError(f"Model type {self.config.model_type} not supported.")
        init_model_func = lvu_init_model_map[self.config.model_type]
        run_model_func = lvu_run_model_map[self.config.model_type]
        model = init_model_func(self.model, self.config)
        self.run_model_func = run_model_func.__get__(self)
        if self.config.model_type in lvu_chat_model_map:
            self.chat_model_func = lvu_chat_model_map[self.config.model_type].__get__(self)
        return model

    def generate(self, question, video_path, **generation_kwargs):
        if self.config.model_type not in lvu_run_model_map:
            raise ValueError(f"Model type {self.config.model_type} not supported.")
        return self.run_model_func(question, video_path, **generation_kwargs)

    def chat(self, messages: dict, **generation_kwargs):
        if self.config.model_type not in lvu_run_model_map:
            raise ValueError(f"Model type {self.config.model_type} not supported.")
        return self.chat_model_func(messages, **generation_kwargs)
