"""Side streams that really run beside the main stream.

ROCm multiplexes HIP streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4 per priority level); two streams that land on the
same queue execute in SUBMISSION order although they are different streams.  Measured on the MI355X box in round 4
(tools/probe/probe_stream_alias.py -> profiles/r4_stream_hw_queue_aliasing.txt): of ten streams from torch's pool, #6 shared a queue
with the default stream and #5 / #9 with #0 — and the pipeline's copy stream happened to share one with its ViT stream (the H2D copy of
group g+1's frames waited for ViT(g) to finish) or, once a hipGraph capture had taken more streams from the pool, with the main stream
(the copy waited for prefill(g): the sequential plugin's ViT then ran strictly AFTER every group, 16 ms of idle LLM stream per group —
VERDICT r3 Weak #7's "ViT-scheduling artefact").  High-priority streams (priority -1) live on queues of their own.

So the pipeline does not take "a stream" and hope: `side_streams` hands out a high-priority copy stream and a ViT stream and VERIFIES on
the device that work on each overtakes a kernel train submitted earlier on the main stream and on the other one (a ~10 ms test per
candidate, once per (device, main stream) and cached)."""
from __future__ import annotations

import sys
from typing import Dict, Tuple

import torch

_CACHE: Dict[Tuple[int, int], Tuple[torch.cuda.Stream, torch.cuda.Stream, dict]] = {}


_TRAIN_REPS: Dict[int, int] = {}


def _train_reps(a: torch.Tensor, stream: torch.cuda.Stream) -> int:
    """GEMM repetitions that keep a stream busy for ~8 ms on this device (calibrated once): long against launch latencies and event
    granularity, so that "finished in under half the train's time" is unambiguous."""
    key = a.device.index or 0
    if key not in _TRAIN_REPS:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            for _ in range(4):
                torch.mm(a, a)
            e0.record(stream)
            for _ in range(32):
                torch.mm(a, a)
            e1.record(stream)
        e1.synchronize()
        per = max(e0.elapsed_time(e1) / 32, 1e-3)
        _TRAIN_REPS[key] = int(min(4000, max(16, 8.0 / per)))
    return _TRAIN_REPS[key]


def _overtakes(first: torch.cuda.Stream, second: torch.cuda.Stream, a: torch.Tensor, probe: torch.Tensor) -> bool:
    """True iff a tiny kernel on `second` finishes well before a kernel train submitted EARLIER on `first` (= separate hardware queues)."""
    with torch.cuda.stream(second):                                     # first use of a stream sets up its queue: not part of the test
        probe.add_(1)
    reps = _train_reps(a, first)
    torch.cuda.synchronize(a.device)
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(first)
    with torch.cuda.stream(first):
        for _ in range(reps):
            torch.mm(a, a)
        ea.record(first)
    with torch.cuda.stream(second):
        probe.add_(1)
        eb.record(second)
    ea.synchronize(); eb.synchronize()
    return e0.elapsed_time(eb) < 0.5 * e0.elapsed_time(ea)


def side_streams(device: torch.device, main: torch.cuda.Stream = None):
    """-> (vit_stream, copy_stream, report).  copy_stream: high priority (frame uploads go ahead of compute and use queues of their own);
    vit_stream: default priority.  Each is checked against the main stream and against the other; candidates that share a hardware
    queue are skipped (they stay in torch's pool, nothing is leaked)."""
    main = main if main is not None else torch.cuda.current_stream(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), main.cuda_stream)
    if key in _CACHE:
        return _CACHE[key]
    a = torch.randn(3072, 3072, device=device, dtype=torch.bfloat16)
    probe = torch.zeros(64, device=device)
    for _ in range(2):                                                  # first-use costs (hipBLASLt plan, module load) out of the timings
        torch.mm(a, a)
    probe.add_(1)
    report = {"copy_candidates_skipped": 0, "vit_candidates_skipped": 0, "verified": True}
    copy = None
    for _ in range(6):
        s = torch.cuda.Stream(device, priority=-1)
        if _overtakes(main, s, a, probe):
            copy = s
            break
        report["copy_candidates_skipped"] += 1
    vit = None
    for _ in range(16):
        s = torch.cuda.Stream(device)
        if _overtakes(main, s, a, probe) and (copy is None or (_overtakes(s, copy, a, probe) and _overtakes(copy, s, a, probe))):
            vit = s
            break
        report["vit_candidates_skipped"] += 1
    if copy is None or vit is None:
        report["verified"] = False
        print("[quickprefill] could not find side streams on hardware queues of their own (GPU_MAX_HW_QUEUES too small?): frame upload / ViT "
              "may serialise with the LLM stream", file=sys.stderr, flush=True)
        copy = copy or torch.cuda.Stream(device, priority=-1)
        vit = vit or torch.cuda.Stream(device)
    torch.cuda.synchronize(device)
    _CACHE[key] = (vit, copy, report)
    return _CACHE[key]
