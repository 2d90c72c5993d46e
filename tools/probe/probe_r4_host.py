"""Round-4 GPU-box probes: (1) how many CPUs this container may really use (affinity, cgroup quota); (2) why the rocprofv3 --pmc child
of bench.py failed; (3) per-group event timeline of the sequential vs overlapped plugin on cfg4s (when does ViT(g+1) start relative to
prefill(g)?)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = {}
out["cpu_count"] = os.cpu_count()
out["affinity"] = len(os.sched_getaffinity(0))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpu.stat"):
    try:
        out[f] = open(f).read().strip()[:300]
    except Exception as e:
        out[f] = f"{type(e).__name__}"
out["loadavg"] = open("/proc/loadavg").read().strip()
print(json.dumps(out, indent=1), flush=True)

which = sys.argv[1:] or ["pmc", "timeline"]
if "pmc" in which:
    import shutil, tempfile
    tmp = tempfile.mkdtemp(prefix="qp_pmc_dbg_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", tmp, "-o", "pmc_FETCH_SIZE", "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--config", "cfg4s", "--window", "22:24", "--lean", "--no-kernel-timing", "--steps", "5", "--warmup", "1"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", QP_BENCH_NO_PMC="1"), capture_output=True, text=True, timeout=600)
    print("PMC child rc", r.returncode)
    print("STDOUT head:", r.stdout[:1500])
    err = r.stderr
    i = err.find("Traceback")
    print("STDERR:", err[i - 500:i + 3000] if i >= 0 else err[:3000])
    import glob
    print(sorted(glob.glob(os.path.join(tmp, "**", "*"), recursive=True))[:20])
    shutil.rmtree(tmp, ignore_errors=True)

if "timeline" in which:
    import torch
    import bench
    from quickvideo_amd.pipeline import PrefillPipeline
    args = type("A", (), dict(steps=5, warmup=1, window=None))()
    dev = torch.device("cuda", 0)
    res, eng, ctx = bench.measure(args, "cfg4s", dev, 0, 1, "single", (1, 1), None, timing="off")
    # monkeypatch the breakdown to dump the raw timeline
    dumps = {}
    orig = PrefillPipeline._device_breakdown

    def spy(tm, origin, trace):
        at = lambda e: round(origin.elapsed_time(e), 2)
        dumps[len(dumps)] = [(at(h), at(v0), at(v1), at(p0), at(p1)) for h, v0, v1, p0, p1 in trace]
        return orig(tm, origin, trace)
    PrefillPipeline._device_breakdown = staticmethod(spy)
    if "decode_first" in which:                      # what bench.py does before its pipeline leg: a captured hipGraph decode + big allocations
        print("decode leg:", bench.decode_leg(eng, 1))
        print("peaked leg:", list(bench.peaked_attention_leg(eng.ops, dev)["by_score_sigma"].items())[:1])
    r = bench.pipeline_leg("cfg4s", eng, dev, vit_alone=("vit_alone" in which))
    for k, tl in dumps.items():
        print("run", k, "(h2d, vit0, vit1, pre0, pre1) ms for groups 10..16:")
        for row in tl[10:17]:
            print("   ", row)
    print({m: (r[m]["ttft_ms"], r[m]["gpu"]["main_stream_gap_before_group_ms"]["p50"]) for m in ("overlapped", "sequential")})
