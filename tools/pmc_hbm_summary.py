"""Condense tools/pmc_hbm_kernels.sh output: per HBM-bound kernel, the steady-state launches' duration (kernel trace), TCC FETCH_SIZE /
WRITE_SIZE (KB; gfx950: FETCH_SIZE counts half the bytes of wide coalesced loads -> x2, MI355X_MICROARCH.md), achieved GB/s against the
8 TB/s peak and the algorithmic bytes of the launch (SURVEY 8d / DESIGN 3)."""
import collections, csv, glob, json, os, sys
out, tag, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
dst = os.path.join(out, "summary"); os.makedirs(dst, exist_ok=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
model, frames, fh, fw, gs, rho, prefix, tail = bench.CONFIGS[cfg]
spec = bench.PRESETS[model]
n = (gs // 2) * (fh // 28) * (fw // 28)
k = int(n * rho)
d, I, hq, hkv, D = spec.hidden, spec.intermediate, spec.n_heads, spec.n_kv_heads, spec.head_dim
KERNELS = {   # name fragment -> (what, algorithmic bytes of one steady-state launch)
    "prune_keys_kernel": ("radix select on the 16-bit norm keys + KV gather/compact (one launch)", 2 * n + 2 * (k * hkv * D * 2 * 2) + 4 * k),
    "rope_append_kernel<true>": ("M-RoPE + K/V append to staging + cross-head key-norm reduction -> 16-bit keys",
                                 n * (hq + 2 * hkv) * D * 2 + n * (hq + 2 * hkv) * D * 2 + 2 * n + n * D * 2),
    "add_rmsnorm_kernel": ("residual add + RMSNorm", 4 * n * d * 2),
    "swiglu_kernel": ("SiLU(gate) * up", 3 * n * I * 2),
}
short = lambda s: s.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def find(pat):
    fs = sorted(glob.glob(os.path.join(out, "**", pat), recursive=True))
    return fs[-1] if fs else None


dur = collections.defaultdict(list)
tr = find("trace_kernel_trace.csv")
for r in csv.DictReader(open(tr)):
    dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = find(f"pmc_{c}_counter_collection.csv")
    acc = collections.defaultdict(list)
    if p:
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == c:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    cnt[c] = acc
res = {"workload": bench.describe(cfg), "n_new_tokens": n, "k_kept": k,
       "note": "durations: rocprofv3 --kernel-trace (no counters); FETCH_SIZE / WRITE_SIZE: separate --pmc passes, KB; read bytes = 2 x FETCH_SIZE x 1024 "
               "(gfx950 counts half of wide coalesced loads, MI355X_MICROARCH.md), write bytes = WRITE_SIZE x 1024; per kernel the median over its "
               "launches (group 0 carries 15 more rows; the prompt tail's launches are tiny and fall outside the median)",
       "peak_gb_s": 8000.0, "kernels": {}}
med = lambda v: sorted(v)[len(v) // 2] if v else None
for frag, (what, alg) in KERNELS.items():
    names = [kname for kname in dur if frag in kname]
    if not names:
        continue
    kn = names[0]
    t_ns = med(dur[kn])
    f_kb, w_kb = med(cnt["FETCH_SIZE"].get(kn, [])), med(cnt["WRITE_SIZE"].get(kn, []))
    e = {"what": what, "launches": len(dur[kn]), "median_us": round(t_ns / 1e3, 2), "algorithmic_bytes": alg,
         "algorithmic_gb_s": round(alg / t_ns, 1), "frac_of_peak_algorithmic": round(alg / t_ns / 8000.0, 4)}
    if f_kb is not None and w_kb is not None:
        tb = 2 * f_kb * 1024 + w_kb * 1024
        e.update({"fetch_size_kb": f_kb, "write_size_kb": w_kb, "traffic_bytes": tb, "traffic_over_algorithmic": round(tb / alg, 3),
                  "traffic_gb_s": round(tb / t_ns, 1)})
    res["kernels"][kn] = e
json.dump(res, open(os.path.join(dst, f"{tag}_hbm_kernels.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
