"""How peaked is the softmax of the BENCHMARK's own workload?  (DESIGN 3.1 said "nearly uniform, sigma ~ 0.05"; checked in round 5.)

Runs the first groups of cfg4s (the 1-hour video's first 6 minutes: same weights, same inputs, same group size as the metric's
workload) through the per-operator loop and, at chosen layers of the LAST group run, takes the scores the attention kernel is about to
see — q.k / sqrt(D) of 256 sampled query rows of q head 0 against every visible key of kv head 0 (prefix + causal part) — and prints
their standard deviation per query row (mean over the sampled rows) and the mean of the row maximum minus the row mean (how far the
largest score stands out).  One JSON line.   usage: QP_NATIVE_SEGMENT=0 python tools/probe/probe_score_sigma.py [groups=8]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["QP_NATIVE_SEGMENT"] = "0"
import bench  # noqa: E402

G_RUN = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
spec, cfg, plan, eng, embeds, pos, T = bench.build_workload("cfg4s", dev, 0, 1)
starts = bench.group_starts(plan)
stats, state = {}, {"layer": 0, "on": False}
real = eng.ops


class Spy:
    def __getattr__(self, name):
        return getattr(real, name)

    def prefill_attn(self, q, k_prefix, v_prefix, prefix_head_stride, prefix_len, k_new, v_new, new_head_stride, n, n_q, n_kv, head_dim, scale, out,
                     q_row0=0, nq=None):
        l = state["layer"]; state["layer"] += 1
        if state["on"] and l in (0, 1, 7, 13, 20, 27):
            D = head_dim
            rows = torch.arange(0, n, max(1, n // 256), device=q.device)[:256]
            qs = q[rows, 0].float()                                                     # [r, D] q head 0
            kp = k_prefix[0, :prefix_len].float() if prefix_len else torch.empty(0, D, device=q.device)
            kn = k_new[0, :n].float()
            sc = torch.cat([qs @ kp.T, qs @ kn.T], 1) * scale                           # [r, P + n]
            vis = torch.arange(prefix_len + n, device=q.device)[None, :] <= (rows[:, None] + prefix_len)
            sc = sc.masked_fill(~vis, float("nan"))
            mean = torch.nanmean(sc, 1, keepdim=True)
            std = torch.sqrt(torch.nanmean((sc - mean) ** 2, 1))
            mx = torch.where(vis, sc, torch.full_like(sc, -1e30)).max(1).values
            stats[f"layer{l}"] = {"score_sigma_per_query": round(float(std.mean()), 3), "row_max_minus_mean": round(float((mx - mean[:, 0]).mean()), 3),
                                  "visible_keys": int(prefix_len + n), "q_std": round(float(q.float().std()), 3), "k_std": round(float(kn.std()), 3)}
        return real.prefill_attn(q, k_prefix, v_prefix, prefix_head_stride, prefix_len, k_new, v_new, new_head_stride, n, n_q, n_kv, head_dim, scale,
                                 out, q_row0=q_row0, nq=nq)


eng.ops = Spy()
for g in range(G_RUN):
    state["layer"], state["on"] = 0, g == G_RUN - 1
    bench.run_groups(eng, plan, starts, embeds, pos, g, g + 1)
torch.cuda.synchronize()
print(json.dumps({"workload": bench.describe("cfg4s"), "group": G_RUN - 1, "what": "scores q.k/sqrt(D) of q head 0 vs kv head 0, 256 sampled query rows, "
                  "all visible keys (pruned prefix + causal part): std per query row (mean over rows), row max - row mean", "by_layer": stats}))
