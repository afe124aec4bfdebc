#!/usr/bin/env python3
"""Do two independent branches of ONE captured HIP graph run concurrently on this runtime?  Two chains of small dependent
kernels (latency-bound, like the collate of a molecule batch next to the step of the previous one), captured (a) one
after the other on one stream, (b) forked onto a side stream and joined.  Prints the replay time of each form."""
import torch
dev = torch.device("cuda:0")
K = 40
x = torch.randn(1 << 14, device=dev); y = torch.randn(1 << 14, device=dev)


def chain(t):
    for _ in range(K):
        t.mul_(1.0001)


def capture(fork):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain(x); chain(y)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        if fork:
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                chain(y)
            chain(x)
            cur.wait_stream(side)
        else:
            chain(x); chain(y)
    return g


def timeit(g, n=200):
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


gs, gf = capture(False), capture(True)
for r in range(3):
    print(f"serial chain of 2 x {K} kernels: {timeit(gs):8.1f} us   forked + joined: {timeit(gf):8.1f} us")

# (c) two graphs, one per chain, replayed on two streams at the same time (host issues both; an event pair per
#     iteration keeps iteration i + 1 of each stream behind iteration i of the other: the double-buffer dependency)
def cap1(t):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(t)
    return g


gx, gy = cap1(x), cap1(y)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def two_streams(n=200, sync_each=True):
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(sa): gx.replay()
        with torch.cuda.stream(sb): gy.replay()
        if sync_each:
            sa.wait_stream(sb); sb.wait_stream(sa)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def one_stream(n=200):
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(n):
        gx.replay(); gy.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for r in range(3):
    print(f"two graphs on one stream: {one_stream():8.1f} us   on two streams, joined per iteration: {two_streams():8.1f} us"
          f"   two streams, free-running: {two_streams(sync_each=False):8.1f} us")
