"""How many INDEPENDENT hardware queues does a process get, as a function of GPU_MAX_HW_QUEUES?

A multi-GPU rank of the pipeline owns more HIP streams than the runtime's default of 4 hardware queues: the LLM stream, the ViT stream,
the copy stream (rank 0) and one RCCL stream per process group it talks on (front end, pair-send, pair-recv, stage group).  Streams that
land on one hardware queue run in submission order; an RCCL kernel that spins for its peer then holds back whatever shares its queue
(streams.py found exactly this for the copy stream).  This probe partitions N streams — default-priority ones and high-priority ones
like torch's RCCL streams — into alias classes: a ~100 ms spin on stream i, then a tiny kernel on every other stream j; j is in i's class
iff its kernel finishes after the spin.

    python tools/probe/probe_hw_queues.py            # runs itself under GPU_MAX_HW_QUEUES = unset, 4, 8, 16
"""
import json
import os
import subprocess
import sys


def classes(n_default=8, n_high=6):
    import torch
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(n_default)] + [torch.cuda.Stream(dev, priority=-1) for _ in range(n_high)]
    names = ["main"] + [f"d{i}" for i in range(n_default)] + [f"h{i}" for i in range(n_high)]
    x = [torch.zeros(1024, device=dev) for _ in streams]
    # warm every stream (first launches load code objects), then calibrate the spin to ~100 ms on a warm kernel
    for j, sj in enumerate(streams):
        with torch.cuda.stream(sj):
            x[j].add_(1)
    cyc = 20_000_000
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.cuda._sleep(cyc); e1.record(); torch.cuda.synchronize()
        cyc = max(1_000_000, int(cyc * 100.0 / max(e0.elapsed_time(e1), 1e-3)))
    parent = list(range(len(streams)))

    def find(a):
        while parent[a] != a:
            a = parent[a]
        return a

    for i, si in enumerate(streams):
        torch.cuda.synchronize()
        t0, tspin = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(si)
        with torch.cuda.stream(si):
            torch.cuda._sleep(cyc)
            tspin.record(si)
        done = []
        for j, sj in enumerate(streams):
            if j == i:
                done.append(None)
                continue
            with torch.cuda.stream(sj):
                x[j].add_(1)
                e = torch.cuda.Event(enable_timing=True); e.record(sj)
                done.append(e)
        torch.cuda.synchronize()
        spin = t0.elapsed_time(tspin)
        for j, e in enumerate(done):
            if e is not None and t0.elapsed_time(e) > 0.7 * spin:        # (no cross-stream wait on t0: an event wait can ride on the spin's packet)
                parent[find(j)] = find(i)
    groups = {}
    for i, nm in enumerate(names):
        groups.setdefault(find(i), []).append(nm)
    return {"spin_ms": round(spin, 1), "classes": sorted(groups.values(), key=lambda g: names.index(g[0]))}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        print(json.dumps(classes()))
        sys.exit(0)
    for q in (None, "4", "8", "16", "24"):
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)
        if q:
            env["GPU_MAX_HW_QUEUES"] = q
        out = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True, timeout=300)
        try:
            cl = json.loads(out.stdout.strip().splitlines()[-1])
            print(f"GPU_MAX_HW_QUEUES={q or 'unset'}: {len(cl['classes'])} independent classes among main + 8 default + 6 high-priority streams "
                  f"(spin {cl['spin_ms']} ms): {cl['classes']}", flush=True)
        except Exception:
            print(f"GPU_MAX_HW_QUEUES={q or 'unset'}: failed rc={out.returncode}: {out.stderr[-400:]}", flush=True)
