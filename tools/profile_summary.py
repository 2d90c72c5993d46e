"""Condense rocprofv3 output of tools/profile_bench.sh into small committed summaries (profiles/)."""
import csv, json, sys, os, collections
out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "summary"); os.makedirs(dst, exist_ok=True)
rows = list(csv.DictReader(open(os.path.join(out, "trace_kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_cfg2_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-pipeline --no-ttft (cfg2, 1x MI355X)\n")
    f.write("name,calls,total_ns,avg_ns,pct,min_ns,max_ns\n")
    for r in rows:
        f.write(f"\"{r['Name'][:100]}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(out, f"pmc_{c}_counter_collection.csv")
    if not os.path.exists(p): continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    res[c] = {k: {"kb_total": v[0], "launches": v[1], "kb_per_launch": v[0] / v[1]} for k, v in acc.items() if v[0] > 0}
# the attention launch = both forms of attn_fwd_kernel_s6 (4-/8-wave workgroups) + the kv-split combine kernel
att = [k for k in res.get("FETCH_SIZE", {}) if "attn_fwd_kernel_s6" in k]
comb = [k for k in res.get("FETCH_SIZE", {}) if "attn_combine_kernel<128" in k]
summary = {"note": "TCC counters via rocprofv3 --pmc, one counter per pass; FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads 1/2 of "
                   "the bytes of wide (16 B/lane) coalesced streams (MI355X_MICROARCH.md HBM section) -> read bytes = 2 x FETCH_SIZE x 1024",
           "counters": res}
if att:
    launches = sum(res["FETCH_SIZE"][k]["launches"] for k in att)
    fb = sum(res["FETCH_SIZE"][k]["kb_total"] for k in att + comb) * 1024 * 2 / launches
    wb = sum(res.get("WRITE_SIZE", {}).get(k, {"kb_total": 0})["kb_total"] for k in att + comb) * 1024 / launches
    summary["attn_fwd_kernel_s6"] = {"launches": launches, "kernels": att + comb, "hbm_read_bytes_per_launch_corrected": fb,
                                     "hbm_write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb}
json.dump(summary, open(os.path.join(dst, f"{tag}_cfg2_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(summary.get("attn_fwd_kernel_s6")))
