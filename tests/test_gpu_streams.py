"""The pipeline's side streams must run BESIDE the LLM stream, not behind it: ROCm maps HIP streams onto a few hardware queues, and two
streams on one queue execute in submission order (round 4: the copy stream shared a queue with the ViT / main stream and the next group's
frames arrived only after the current group had finished)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _finishes_first(first, second, work_second):
    """Submit a ~25 ms GEMM train on `first`, THEN `work_second` on `second`; True iff the second finishes in under half the train's time."""
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    torch.mm(a, a); torch.cuda.synchronize()
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(first)
    with torch.cuda.stream(first):
        for _ in range(30):
            torch.mm(a, a)
        ea.record(first)
    with torch.cuda.stream(second):
        work_second()
        eb.record(second)
    torch.cuda.synchronize()
    return e0.elapsed_time(eb) < 0.5 * e0.elapsed_time(ea), e0.elapsed_time(ea), e0.elapsed_time(eb)


def test_side_streams_are_on_hardware_queues_of_their_own():
    from quickvideo_amd.streams import side_streams
    dev = torch.device("cuda", 0)
    # take a few streams out of torch's pool first, as an application (or a hipGraph capture) would: the round-robin position must not matter
    _ = [torch.cuda.Stream(dev) for _ in range(5)]
    main = torch.cuda.current_stream(dev)
    vit, copy, report = side_streams(dev)
    assert report["verified"], report
    assert copy.priority < vit.priority or copy.priority == -1
    host = torch.empty(32 << 20, dtype=torch.uint8).pin_memory()
    dst = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
    x = torch.zeros(1 << 20, device=dev)
    up = lambda: dst.copy_(host, non_blocking=True)
    k = lambda: x.add_(1)
    for first, second, work, what in ((main, copy, up, "frame upload behind the LLM stream"), (vit, copy, up, "frame upload behind the ViT stream"),
                                      (main, vit, k, "ViT behind the LLM stream"), (copy, vit, k, "ViT behind the copy stream")):
        ok, ta, tb = _finishes_first(first, second, work)
        assert ok, f"{what}: second stream finished at {tb:.1f} ms, the train before it at {ta:.1f} ms — the two share a hardware queue"
    assert side_streams(dev)[0] is vit                                      # cached per (device, main stream)
