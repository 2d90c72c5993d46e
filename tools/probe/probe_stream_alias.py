"""Do two HIP streams share a hardware queue?  ROCm multiplexes HIP streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4);
streams that land on the same queue execute in SUBMISSION order although they are different streams.  Round 4 found the pipeline's copy
stream aliased with the ViT stream (and, after a hipGraph capture had created more streams, with the main stream): the H2D copy of group
g+1's frames then waited for ViT(g) / prefill(g) to finish.  This probe measures aliasing directly: a ~40 ms kernel train on stream A,
then a small H2D copy on stream B; B is independent iff its copy finishes long before A's train."""
import os, sys, time, torch
dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
host = torch.empty(16 << 20, dtype=torch.uint8).pin_memory()
dst = torch.empty(16 << 20, dtype=torch.uint8, device=dev)


def independent(sa, sb, kind="copy"):
    """True if work on sb overtakes a long kernel train submitted earlier on sa."""
    torch.cuda.synchronize()
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(sa)
    with torch.cuda.stream(sa):
        for _ in range(40):
            a @ a
        ea.record(sa)
    with torch.cuda.stream(sb):
        if kind == "copy":
            dst.copy_(host, non_blocking=True)
        else:
            dst.add_(1)
        eb.record(sb)
    torch.cuda.synchronize()
    ta, tb = e0.elapsed_time(ea), e0.elapsed_time(eb)
    return tb < 0.5 * ta, round(ta, 1), round(tb, 1)


main = torch.cuda.current_stream(dev)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
streams = [torch.cuda.Stream(dev) for _ in range(10)]
for i, s in enumerate(streams):
    print(f"default-priority stream #{i}: vs main copy {independent(main, s)}, kernel {independent(main, s, 'k')}; vs stream#0 copy {independent(streams[0], s) if i else '-'}")
for pr in (-1, 1, 2):
    try:
        s = torch.cuda.Stream(dev, priority=pr)
        print(f"priority {pr} stream (reports priority {s.priority}): vs main copy {independent(main, s)}, kernel {independent(main, s, 'k')}; vs stream#0 {independent(streams[0], s)}; vs stream#1 {independent(streams[1], s)}")
        s2 = torch.cuda.Stream(dev, priority=pr)
        print(f"   second priority {pr} stream vs the first: {independent(s, s2)}")
    except Exception as e:
        print(f"priority {pr}: {type(e).__name__}: {e}")
# after a graph capture
g = torch.cuda.CUDAGraph()
x = torch.zeros(1024, device=dev)
side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        x.add_(1)
g.replay(); torch.cuda.synchronize()
print("after a graph capture + replay:")
for i, s in enumerate(streams[:4]):
    print(f"  stream #{i}: vs main copy {independent(main, s)}; vs stream#0 {independent(streams[0], s) if i else '-'}")
