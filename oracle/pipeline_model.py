"""The tiny Qwen2-VL / Qwen2.5-VL checkpoints of GV11 (tests/golden/gv11_e2e_pipeline.*), re-created from a seed with the INSTALLED
transformers — by oracle/make_golden.py when it generates the fixture and by the tests when they need the same weights (the checkpoints are not
stored; `weights_sha256` in the fixture pins them).  Test infrastructure only: nothing under quickvideo_amd/ imports this.  Does not touch
/root/reference."""
import hashlib

import torch

PIPE_TEXT = dict(hidden_size=256, num_attention_heads=2, num_key_value_heads=1, intermediate_size=512, num_hidden_layers=2, vocab_size=320,
                 rms_norm_eps=1e-6, tie_word_embeddings=False, rope_parameters=dict(rope_type="default", mrope_section=[16, 24, 24], rope_theta=1_000_000.0))
PIPE_IDS = dict(video_token_id=300, vision_start_token_id=301, vision_end_token_id=302)
PIPE_VISION = {
    "qwen2-vl": dict(depth=2, embed_dim=64, hidden_size=256, num_heads=4, mlp_ratio=2, patch_size=14, spatial_merge_size=2, temporal_patch_size=2),
    "qwen2.5-vl": dict(depth=2, hidden_size=64, intermediate_size=80, num_heads=4, out_hidden_size=256, window_size=112, fullatt_block_indexes=[1],
                       patch_size=14, spatial_merge_size=2, temporal_patch_size=2, tokens_per_second=2),
}
PIPE_FRAMES = dict(n=8, h=112, w=168, seed=5, fps=2.0)          # a .npy "video": uint8 [8, 3, 112, 168], served at 2 fps; num_frames = 8 samples them all
PIPE_QUESTION = "what is this"
PIPE_GROUP = 4                                                  # video_group_size -> 2 groups of 2 temporal patches
PIPE_DECODE = 4
PIPE_CASES = [("qwen2-vl", 1.0), ("qwen2-vl", 0.5), ("qwen2.5-vl", 1.0), ("qwen2.5-vl", 0.5)]
CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)


PIPE_SEED = {"qwen2-vl": 0, "qwen2.5-vl": 5}       # init seeds whose 4 greedy steps are decided by >= 0.4 logit (6x the bf16-vs-fp32 distance of the oracle itself)


def build_hf_pipeline_model(family: str, seed=None):
    """The tiny checkpoint of GV11 (also called by the GPU test, from the installed transformers): seeded init, every parameter rounded to
    a bf16-representable value (the engine holds bf16 weights; the oracle runs the same numbers in fp32 and in bf16)."""
    if family == "qwen2.5-vl":
        from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLConfig as Cfg
        from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VLForConditionalGeneration as Cls
    else:
        from transformers import Qwen2VLConfig as Cfg, Qwen2VLForConditionalGeneration as Cls
    cfg = Cfg(text_config=dict(PIPE_TEXT), vision_config=dict(PIPE_VISION[family]), **PIPE_IDS)
    cfg._attn_implementation = "eager"
    torch.manual_seed(PIPE_SEED[family] if seed is None else seed)
    hf = Cls(cfg).eval()
    with torch.no_grad():
        for p_ in hf.parameters():
            p_.copy_(p_.to(torch.bfloat16).float())
        # seeded inits leave lm_head / embeddings at std 0.02: first-token margins of ~1e-3, below any bf16 tolerance.  Scale the head so
        # that greedy decoding is decided by the model, not by rounding (logits |x| <= ~6, like GV5's synthetic head)
        hf.lm_head.weight.mul_(16.0)
        hf.model.language_model.embed_tokens.weight.mul_(16.0)
    return hf, cfg


def state_sha(model) -> str:
    h = hashlib.sha256()
    for k, v in sorted(model.state_dict().items()):
        h.update(k.encode()); h.update(v.detach().to(torch.float32).contiguous().numpy().tobytes())
    return h.hexdigest()


