"""TEST INFRASTRUCTURE (not product code): distance of the CPU oracle from the GV8 full-depth fixture.

GV8 (tests/golden/gv8_deep.*) is the reference composite — installed transformers' Qwen2-VL decoder (sdpa) + the reference's
own post_process_kv_cache (lvu/utils.py:197-376) after every attention — at 28 layers and the Qwen2-VL-7B dims.  This script
runs the *independent* CPU restatement (oracle/qp_oracle.py::group_prefill) on the same hash-generated weights and input rows
and records how far an independent bf16 implementation lands from the fixture: first-token logits (max|d|, cosine, argmax) and
the kept-set overlap per layer (min over groups).  tests/test_gpu_deep.py derives its bars from this record (<= 1.25x the
oracle's own logit distance; overlap floor = the oracle's per-layer overlap minus a stated margin), so the GPU path is held to
"as close to the reference as an independent CPU implementation is", not to whatever the GPU happened to produce.

    python oracle/calibrate_deep.py            # ~10-25 min of CPU, ~35 GB of RAM; writes tests/golden/gv8_deep_oracle_calibration.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import qp_oracle as O  # noqa: E402


def main():
    gdir = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gdir, "gv8_deep.json")))
    gold = np.load(os.path.join(gdir, "gv8_deep.npz"))
    spec = O.TextSpec(**meta["spec"])
    torch.set_num_threads(os.cpu_count())
    t0 = time.time()
    w = O.hashed_text_weights(spec, seed=meta["weight_seed"])
    T = meta["prefix"] + (meta["frames"] // 2) * (meta["grid_h"] // 2) * (meta["grid_w"] // 2) + meta["tail"]
    plan = O.plan_groups(meta["frames"], meta["group_size"], meta["grid_h"], meta["grid_w"], meta["prefix"], T)
    assert list(plan.tokens) == meta["group_tokens"]
    pos, _ = O.mrope_positions(meta["prefix"], (meta["frames"] // 2, meta["grid_h"], meta["grid_w"]), meta["tail"])
    embeds = O.hashed_normal((T, spec.hidden), meta["embed_seed"], 0.5)
    print(f"weights + inputs in {time.time() - t0:.0f} s", flush=True)
    with torch.no_grad():
        out = O.group_prefill(w, spec, embeds, pos, plan.tokens, O.PruneCfg(top_p=meta["top_p"]))
    L, G = spec.n_layers, len(plan.tokens)
    assert out["cache_len"] == [int(x) for x in gold["cache_len"]], "cache lengths differ from the fixture"
    overlap = np.zeros((G, L))
    for gi in range(G):
        for l in range(L):
            got, want = out["kept"][gi][l], gold[f"kept_g{gi}_l{l}"].astype(np.int64)
            assert len(got) == len(want)
            overlap[gi, l] = len(set(got.tolist()) & set(want.tolist())) / len(want)
    logits, ref = out["logits"].numpy(), gold["logits"]
    rec = {
        "what": "oracle/qp_oracle.py::group_prefill (torch-CPU bf16) vs tests/golden/gv8_deep.* (reference composite)",
        "made_by": "oracle/calibrate_deep.py",
        "torch": torch.__version__, "threads": torch.get_num_threads(),
        "cache_len_equal": True,
        "logits_max_abs_diff": float(np.max(np.abs(logits - ref))),
        "logits_cosine": float(np.dot(logits, ref) / (np.linalg.norm(logits) * np.linalg.norm(ref))),
        "logit_absmax": float(np.abs(ref).max()),
        "argmax": int(np.argmax(logits)), "reference_argmax": int(meta["argmax"]),
        "overlap_min_over_groups_by_layer": [round(float(x), 6) for x in overlap.min(axis=0)],
        "overlap_by_group_layer": [[round(float(x), 6) for x in row] for row in overlap],
        "seconds": round(time.time() - t0, 1),
    }
    dst = os.environ.get("QP_GOLDEN_OUT", gdir)
    with open(os.path.join(dst, "gv8_deep_oracle_calibration.json"), "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
    print(json.dumps({k: v for k, v in rec.items() if k != "overlap_by_group_layer"}, indent=1))


if __name__ == "__main__":
    main()
