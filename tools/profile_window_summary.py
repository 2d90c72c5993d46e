"""Condense the rocprofv3 output of tools/profile_cfg4_window.sh into small committed summaries (profiles/)."""
import collections, csv, glob, json, os, sys
out, tag, win = sys.argv[1], sys.argv[2], sys.argv[3]
dst = os.path.join(out, "summary"); os.makedirs(dst, exist_ok=True)
g0, g1 = (int(v) for v in win.split(":"))

def find(pat):
    fs = sorted(glob.glob(os.path.join(out, "**", pat), recursive=True))
    return fs[-1] if fs else None

short = lambda k: k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
st = find("trace_kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    with open(os.path.join(dst, f"{tag}_cfg4_window_kernel_stats.csv"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --window {win} --steps 20 --warmup 1 --lean   (cfg4 = the 1-hour video; "
                f"groups [{g0},{g1}) from a fast-forwarded arena + the 4-group warm-up; 1x MI355X)\n")
        f.write("name,calls,total_ns,avg_ns,pct,min_ns,max_ns\n")
        for r in rows:
            f.write(f"\"{short(r['Name'])}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
# per-dispatch trace: steady-state attention launches are the long ones (warm-up launches run over short prefixes)
tr = find("trace_kernel_trace.csv")
att_ns = []
if tr:
    for r in csv.DictReader(open(tr)):
        if "attn_fwd_kernel_s6" in r["Kernel_Name"]:
            att_ns.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
steady = sorted(att_ns)[-(g1 - g0) * 28:] if att_ns else []
# everything launched after the arena fast-forward (the last long RNG fill) belongs to the window: per-kernel breakdown of the window only
breakdown = None
if tr:
    disp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(tr))]
    disp.sort()
    fills = [i for i, d in enumerate(disp) if "distribution_elementwise" in d[2] and d[1] - d[0] > 200_000]
    if fills:
        win_d = disp[fills[-1] + 1:]
        agg = collections.defaultdict(lambda: [0, 0])
        for a, b, k in win_d:
            agg[k][0] += b - a; agg[k][1] += 1
        span = win_d[-1][1] - win_d[0][0]
        busy = sum(v[0] for v in agg.values())
        breakdown = {"window_span_ms": span / 1e6, "gpu_busy_ms": busy / 1e6, "groups": g1 - g0,
                     "kernels": [{"name": k, "calls": v[1], "total_ms": round(v[0] / 1e6, 3), "avg_us": round(v[0] / v[1] / 1e3, 2),
                                  "pct_of_busy": round(100.0 * v[0] / busy, 2)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:24]]}
res = {}
for p in sorted(glob.glob(os.path.join(out, "**", "pmc_*_counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        acc[r["Counter_Name"]][short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for cn, d in acc.items():
        res[cn] = {k: v for k, v in d.items()}
def steady_mean(counter, kern_pat):
    vals = []
    for k, v in res.get(counter, {}).items():
        if kern_pat in k:
            vals += v
    vals = sorted(vals)[-(g1 - g0) * 28:]              # the window's launches carry the largest counts
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
n, hq, hkv = 2240, 28, 4
ks = [1120] * 450; ks[0] = 1127
P_mid = sum(ks[:(g0 + g1) // 2])
alg_bytes = (P_mid + n) * hkv * 128 * 2 * 2 + 2 * n * hq * 128 * 2
alg_flops = 4 * hq * 128 * (n * P_mid + n * (n + 1) / 2)
summary = {"window": f"cfg4 groups [{g0},{g1}): n=2240 new tokens over ~{P_mid} pruned prefix rows per layer", "attention_launches_in_window": len(steady),
           "note": "TCC counters via rocprofv3 --pmc, one counter group per pass; FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads 1/2 of "
                   "the bytes of wide (16 B/lane) coalesced streams (MI355X_MICROARCH.md, HBM section) -> read bytes = 2 x FETCH_SIZE x 1024; "
                   "WRITE_SIZE taken as is (uncalibrated)"}
if breakdown:
    summary["window_kernel_breakdown"] = breakdown
if steady:
    avg = sum(steady) / len(steady)
    summary["attn_avg_launch_ms_kernel_trace"] = avg / 1e6
    summary["attn_tflops_kernel_trace"] = alg_flops / (avg * 1e-9) / 1e12
f, nf = steady_mean("FETCH_SIZE", "attn_fwd_kernel_s6")
w, nw = steady_mean("WRITE_SIZE", "attn_fwd_kernel_s6")
if f is not None:
    rd, wr = f * 1024 * 2, (w or 0) * 1024
    summary["attn_fwd_kernel_s6"] = {"launches": nf, "hbm_read_bytes_per_launch_corrected": rd, "hbm_write_bytes_per_launch": wr,
                                     "traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg_bytes,
                                     "window": summary["window"]}
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES",
          "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
    v, nv = steady_mean(c, "attn_fwd_kernel_s6")
    if v is not None:
        summary.setdefault("attn_sq_counters_per_launch", {})[c] = v
sq = summary.get("attn_sq_counters_per_launch", {})
if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "GRBM_GUI_ACTIVE" in sq and steady:
    cyc = sq["GRBM_GUI_ACTIVE"] / 8.0                              # the counter is summed over the 8 XCDs
    n_mfma, alg_mfma = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0, alg_flops / 32768.0
    d = {"note": "GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts MFMA-pipe cycles summed over the 1024 SIMDs (= 32 x "
                 "N_mfma for 32x32x16 bf16); SQ_INSTS_* count wave instructions",
         "shader_clock_ghz_during_profiled_launch": round(cyc / (sum(steady) / len(steady)), 3),
         "mfma_busy_fraction_of_cycles": round(sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4),
         "mfma_instructions": n_mfma, "algorithmic_mfma_instructions": alg_mfma, "mfma_over_algorithmic": round(n_mfma / alg_mfma, 4)}
    if "SQ_INSTS_VALU" in sq:
        d["valu_instructions_per_mfma"] = round((sq["SQ_INSTS_VALU"] - n_mfma) / n_mfma, 3)
    if "SQ_INSTS_LDS" in sq:
        d["lds_instructions_per_mfma"] = round(sq["SQ_INSTS_LDS"] / n_mfma, 3)
    summary["derived"] = d
json.dump(summary, open(os.path.join(dst, f"{tag}_cfg4_window_pmc.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
