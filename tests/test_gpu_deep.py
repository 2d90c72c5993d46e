"""Full-depth parity at the real model dimensions (GV8): 28 decoder layers at the Qwen2-VL-7B dims (d=3584, 28/4 heads,
I=18944; vocab cut to 32768), bf16, 3 video groups x 1280 tokens (+15 prefix) + a 30-token prompt tail, key-norm rho=0.5.

Reference side (oracle/make_golden.py::gen_e2e_deep, run in the build container): installed transformers' Qwen2-VL decoder
(sdpa) + the REFERENCE's post_process_kv_cache hooked after every attention.  Weights and input rows are hash-generated
(oracle.hashed_text_weights): the same bits on the CPU that made the fixture and on the GPU that replays it.

What is asserted, and why it is stated this way: kept INDICES are bit-exact functions of the keys (tests/test_gpu_ops.py), but
the keys of layer l depend on 28 bf16 GEMM chains whose fp32 accumulation order differs between hipBLASLt and the CPU GEMMs,
and a group's bf16 norms fall on a few dozen values — so a rounding flip in one key norm can move a token across the threshold.
The test therefore pins (1) every cache length exactly, (2) the first generated token exactly (reference top-2 margin 0.9),
(3) the first-token logits within DEEP_ATOL / cosine, and (4) the per-layer kept-set overlap with the reference, which must
stay above a bound that is allowed to fall with depth (reported in the assertion message and printed).

Where the bars come from (round 3): NOT from the GPU's own output.  tests/golden/gv8_deep_oracle_calibration.json is the record of
the independent CPU restatement (oracle/qp_oracle.py, torch-CPU bf16) run against the same fixture by oracle/calibrate_deep.py
(13 min of CPU): max|d logit| 0.2129, cosine 0.998940, same argmax, per-layer kept-set overlap 1.000 ... 0.969.  The GPU path
may be at most 1.25x as far from the reference composite as that CPU implementation is:
    max|d| <= 1.25 * 0.2129 = 0.266;   1 - cosine <= 1.25 * (1 - 0.998940);
    overlap(layer l) >= 1 - 1.25 * (1 - e(l)) - 0.005,  e(l) = min over layers <= l of the oracle's overlap (its monotone
    envelope: single layers of either implementation wobble by a few tokens of 640)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import qp_oracle as O
from quickvideo_amd import planner
from quickvideo_amd.engine import QuickPrefillEngine
from quickvideo_amd.lvu_config import LVUConfig
from quickvideo_amd.spec import TextSpec
from quickvideo_amd.weights import DecoderWeights

pytestmark = pytest.mark.gpu

SLACK = 1.25                                   # the GPU may be this much farther from the reference than the CPU oracle is


def bars_from_oracle_calibration(golden_dir):
    cal = json.load(open(os.path.join(golden_dir, "gv8_deep_oracle_calibration.json")))
    assert cal["cache_len_equal"] and cal["argmax"] == cal["reference_argmax"]
    atol = SLACK * cal["logits_max_abs_diff"]
    cos = 1.0 - SLACK * (1.0 - cal["logits_cosine"])
    env = np.minimum.accumulate(np.asarray(cal["overlap_min_over_groups_by_layer"]))
    floor = 1.0 - SLACK * (1.0 - env) - 0.005
    return atol, cos, floor, cal


def test_full_depth_7b_dims_vs_reference_composite(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "gv8_deep.json")))
    gold = np.load(os.path.join(golden_dir, "gv8_deep.npz"))
    so = O.TextSpec(**meta["spec"])
    spec = TextSpec(**meta["spec"])
    w = O.hashed_text_weights(so, seed=meta["weight_seed"], device="cuda")          # 15 GB generated on the GPU in seconds
    T = meta["prefix"] + (meta["frames"] // 2) * (meta["grid_h"] // 2) * (meta["grid_w"] // 2) + meta["tail"]
    plan = planner.plan_groups(meta["frames"], meta["group_size"], meta["grid_h"], meta["grid_w"], meta["prefix"], T)
    assert plan.tokens == meta["group_tokens"] and plan.tail_len == meta["tail_len"]
    pos, delta = planner.mrope_positions(meta["prefix"], (meta["frames"] // 2, meta["grid_h"], meta["grid_w"]), meta["tail"])
    assert int(delta) == meta["rope_delta"]
    embeds = O.hashed_normal((T, spec.hidden), meta["embed_seed"], 0.5, device="cuda")
    dw = DecoderWeights.from_named(spec, w, "cuda:0")
    del w
    eng = QuickPrefillEngine(dw, LVUConfig("x", top_p=meta["top_p"], video_group_size=meta["group_size"]), capacity=T + 8,
                             max_group_tokens=max(plan.tokens + [plan.tail_len]), device="cuda:0")
    eng.kept_trace = []
    post = torch.from_numpy(pos).cuda()
    start = 0
    for n in plan.tokens:
        eng.prefill_group(embeds[start:start + n], post[:, start:start + n])
        start += n
    logits = eng.prefill_tail(embeds[start:], post[:, start:]).cpu().numpy()
    torch.cuda.synchronize()
    L, G = spec.n_layers, len(plan.tokens)
    DEEP_ATOL, DEEP_COS, floor, cal = bars_from_oracle_calibration(golden_dir)
    OVERLAP_FLOOR = lambda layer: floor[layer]                   # noqa: E731
    print(f"bars from the CPU oracle's own distance to the fixture (x{SLACK}): max|d| <= {DEEP_ATOL:.4f}, cosine >= {DEEP_COS:.6f}, "
          f"overlap floor by layer: " + " ".join(f"{x:.3f}" for x in floor))
    # (1) cache lengths: exact
    assert eng.arena.len == list(gold["cache_len"])
    # (4) kept sets per (group, layer)
    kept = [t for t in eng.kept_trace if t[1] is not None]
    assert len(kept) == G * L
    overlap = np.zeros((G, L))
    for gi in range(G):
        for l in range(L):
            layer, idx = kept[gi * L + l]
            assert layer == l
            got, want = idx.cpu().numpy(), gold[f"kept_g{gi}_l{l}"].astype(np.int64)
            assert len(got) == len(want) and np.all(np.diff(got) > 0)
            overlap[gi, l] = len(set(got.tolist()) & set(want.tolist())) / len(want)
    per_layer = overlap.min(axis=0)
    print("kept-set overlap with the reference composite, min over groups, by layer:", " ".join(f"{x:.3f}" for x in per_layer))
    # (2, 3) first token and logits
    ref = gold["logits"]
    err = float(np.max(np.abs(logits - ref)))
    cos = float(np.dot(logits, ref) / (np.linalg.norm(logits) * np.linalg.norm(ref)))
    print(f"first-token logits after 28 layers: max|d| = {err:.4f} (|logit| max {np.abs(ref).max():.2f}), cosine = {cos:.5f}, "
          f"argmax {int(np.argmax(logits))} vs reference {meta['argmax']} (reference top-2 margin {meta['top2_margin']:.3f})")
    assert overlap[:, 0].min() >= 0.99, overlap[:, 0]            # layer 0 sees identical inputs: only GEMM rounding in K
    for l in range(L):
        assert per_layer[l] >= OVERLAP_FLOOR(l), (l, per_layer[l], per_layer)
    assert int(np.argmax(logits)) == meta["argmax"]
    assert err <= DEEP_ATOL and cos >= DEEP_COS, (err, cos)
