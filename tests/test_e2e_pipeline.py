"""GV11: the reference's whole chat_lvu_model (qwen25_lvu_interleaved.py:733-942 / qwen25_lvu.py:538-761) THROUGH THE PLUGIN —
frames -> producer -> (H2D ring) -> patchify -> ViT -> masked_scatter -> group prefill with key-norm pruning -> prompt tail -> first token
-> greedy decode — against the composite oracle's fixture (installed transformers' Qwen2-VL / Qwen2.5-VL + the reference's
post_process_kv_cache, oracle/make_golden.py::gen_e2e_pipeline): cache lengths exact, first-token logits within a stated tolerance,
the four generated token ids equal.  `-m gpu`: lvu.LVU(...).generate() of both plugins (overlapped and sequential) on the HIP path.
Without a GPU the same body runs on the oracle-backed operator double (tests/oracle_ops.py) — that pins the host logic
(planner, prompt ids, positions, scatter, decode positions) here, where the GPU box is not needed to see a mistake."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.pipeline_model import PIPE_CASES, build_hf_pipeline_model, state_sha

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Tolerance (bf16 engine vs the fp32 composite oracle, logits |x| <= ~8): the oracle's OWN bf16 run sits 0.048-0.119 max-abs / cosine
# >= 0.9998 from its fp32 run on these four cases (fixture: bf16_oracle.first_logits_max_abs_diff); the engine must stay within twice
# that distance + 0.02, and within cosine 0.999.  Token ids: the fixture's greedy steps are decided by >= 0.41 logit.
COS_MIN = 0.999


def _fixture():
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "gv11_e2e_pipeline.json")))
    data = np.load(os.path.join(ROOT, "tests", "golden", "gv11_e2e_pipeline.npz"))
    return meta, data


def _checkpoint_dir(tmp_path, family, want_sha):
    from safetensors.torch import save_file
    hf, cfg = build_hf_pipeline_model(family)
    assert state_sha(hf) == want_sha, "the seeded tiny checkpoint differs from the one the fixture was generated with (torch / transformers init changed?)"
    d = tmp_path / family.replace(".", "_")
    d.mkdir(exist_ok=True)
    save_file({k: v.contiguous() for k, v in hf.state_dict().items()}, str(d / "model.safetensors"))
    c = cfg.to_dict()
    c["text_config"]["rope_scaling"] = {"mrope_section": [16, 24, 24]}
    c["text_config"]["rope_theta"] = 1_000_000.0
    json.dump(c, open(d / "config.json", "w"), default=str)
    return str(d)


def _run_case(tmp_path, meta, data, case, model_type, device, ops=None):
    import lvu
    from quickvideo_amd.engine import QuickPrefillEngine
    from quickvideo_amd.lvu import load_native_model
    from tests.test_processor_seam import installed_qwen2vl_processor
    fr = meta["frames"]
    frames = np.random.RandomState(fr["seed"]).randint(0, 256, (fr["n"], 3, fr["h"], fr["w"]), dtype=np.uint8)
    video = str(tmp_path / "v.npy")
    np.save(video, frames)
    model = load_native_model(_checkpoint_dir(tmp_path, case["family"], case["weights_sha256"]), device=device)
    processor = installed_qwen2vl_processor(tuple(case["grid"]))     # the INSTALLED transformers Qwen2VLProcessor (chat template, pad expansion)
    cfg = lvu.LVUConfig(case["family"], model_type=model_type, top_p=case["rho"] if case["rho"] < 1.0 else None,
                        video_group_size=meta["video_group_size"], num_frames=meta["num_frames"])
    obj = lvu.LVU(cfg, model=model, processor=processor)
    if ops is not None:
        obj._ops = ops
    cap, seen = {}, {}
    orig_tail, orig_decode = QuickPrefillEngine.prefill_tail, processor.batch_decode

    def tail(self, e, p):
        cap["logits"] = orig_tail(self, e, p)
        cap["cache_len"] = list(self.arena.len)
        return cap["logits"]

    def batch_decode(seqs, **kw):
        seen["ids"] = [int(t) for t in seqs[0]]
        return orig_decode(seqs, **kw)
    QuickPrefillEngine.prefill_tail, processor.batch_decode = tail, batch_decode
    try:
        text = obj.generate(meta["question"], video, max_new_tokens=meta["decode_steps"], eos_token_id=None)
    finally:
        QuickPrefillEngine.prefill_tail = orig_tail
    assert isinstance(text, list) and len(text) == 1 and isinstance(text[0], str)                 # lvu/lvu.py:45-51 return type
    pipe = obj._pipeline
    P = pipe.plan(__import__("quickvideo_amd.frames", fromlist=["open_video"]).open_video(video), meta["question"])
    assert list(P["prompt"].prefix_ids) == case["prefix_ids"] and list(P["prompt"].tail_ids) == case["tail_ids"]
    assert list(P["plan"].tokens) == case["group_tokens"] and P["plan"].tail_len == case["tail_len"]
    assert cap["cache_len"] == case["cache_len"], (cap["cache_len"], case["cache_len"])            # every layer: pruned prefix + unpruned tail
    ref = data[f"{case['name']}_logits"][0]
    got = cap["logits"].float().cpu().numpy().reshape(-1)
    bar = 2.0 * case["bf16_oracle"]["first_logits_max_abs_diff"] + 0.02
    err = float(np.abs(got - ref).max())
    cos = float(np.dot(got, ref) / (np.linalg.norm(got) * np.linalg.norm(ref)))
    assert err <= bar and cos >= COS_MIN, (case["name"], model_type, err, bar, cos)
    assert seen["ids"] == case["tokens"], (case["name"], model_type, seen["ids"], case["tokens"])
    return err, cos


def test_fixture_is_what_the_generator_describes():
    meta, data = _fixture()
    assert [(c["family"], c["rho"]) for c in meta["cases"]] == [tuple(c) for c in PIPE_CASES]
    for c in meta["cases"]:
        lg = data[f"{c['name']}_logits"]
        assert lg.shape == (meta["decode_steps"], 320) and [int(np.argmax(r)) for r in lg] == c["tokens"]
        assert c["bf16_oracle"]["tokens"] == c["tokens"] and c["bf16_oracle"]["cache_len"] == c["cache_len"]
        assert min(c["margins"]) >= 0.4 > 3 * c["bf16_oracle"]["first_logits_max_abs_diff"]
        if c["rho"] >= 1.0:                                     # pinned to ONE forward of the installed model over the whole prompt
            assert c["composite_vs_whole_forward_max_abs"] < 2e-4 and c["cache_len"] == [len(c["prefix_ids"]) + c["n_video"] + c["tail_len"]] * 2
        else:
            assert c["cache_len"][0] < len(c["prefix_ids"]) + c["n_video"] + c["tail_len"]


@pytest.mark.parametrize("ci", range(len(PIPE_CASES)))
def test_plugin_end_to_end_vs_composite_oracle_on_the_operator_double(tmp_path, ci):
    from tests.oracle_ops import OracleOps
    meta, data = _fixture()
    _run_case(tmp_path, meta, data, meta["cases"][ci], "qwen2vl_mi355x_sequential", "cpu", ops=OracleOps())


@pytest.mark.gpu
@pytest.mark.parametrize("model_type", ["qwen2vl_mi355x", "qwen2vl_mi355x_sequential"])
@pytest.mark.parametrize("ci", range(len(PIPE_CASES)))
def test_plugin_end_to_end_vs_composite_oracle_on_gpu(tmp_path, ci, model_type):
    """a11 on the HIP path: producer thread + pinned ring + copy stream + GPU patchify + HIP ViT + scatter + qp_prefill_segment + tail +
    hipGraph decode, both plugins, Qwen2-VL and Qwen2.5-VL checkpoints, rho in {1, 0.5}."""
    meta, data = _fixture()
    err, cos = _run_case(tmp_path, meta, data, meta["cases"][ci], model_type, "cuda:0")
    print(f"{meta['cases'][ci]['name']} {model_type}: first-token logits max|d| = {err:.4f}, cosine = {cos:.6f}")
