#!/usr/bin/env python3
"""Do the three parts of a skew-plan SpMM (light rows | segmented rows | XCD-pinned rows: disjoint output rows)
overlap when launched on three streams?  RMAT s24, F = 32.  Knob spmm_parts selects the parts of a launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gae_dgl_amd import ops, _lib, workloads as W

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
n = 1 << scale
src, dst = W.rmat_edges(scale, 16, seed=0, device=dev)
ip, ix = ops.csr_from_coo(dst, src, n, n)
del src, dst
plan = ops.spmm_plan(ip, indices=ix, ell=False, n_cols=n)
H = torch.rand(n, F, device=dev)
out = torch.empty(n, F, device=dev)
ref = ops.spmm_raw(ip, ix, H, n, plan=plan).clone()


def knob(v):
    _lib.call("gae_tuning_set", b"spmm_parts", v)


def timed(fn, rounds=6):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rounds


def one():
    ops.spmm_raw(ip, ix, H, n, out=out, plan=plan)


print("one launch, all parts      %.3f ms" % timed(one))
for v, name in ((1, "light rows"), (2, "segmented rows"), (4, "pinned rows")):
    knob(v)
    print("  only %-18s %.3f ms" % (name, timed(one)))
knob(7)
side = [torch.cuda.Stream(), torch.cuda.Stream()]


def three(order=(1, 2, 4)):
    main = torch.cuda.current_stream()
    for st in side:
        st.wait_stream(main)
    knob(order[0]); one()
    for st, v in zip(side, order[1:]):
        with torch.cuda.stream(st):
            knob(v); one()
    for st in side:
        main.wait_stream(st)
    knob(7)


for order in ((1, 2, 4), (4, 2, 1), (2, 4, 1)):
    t = timed(lambda: three(order))
    three(order); torch.cuda.synchronize()
    print("three streams, order %s   %.3f ms   max |diff| %.2e" % (order, t, float((out - ref).abs().max())))


def two():
    main = torch.cuda.current_stream()
    side[0].wait_stream(main)
    knob(1); one()
    with torch.cuda.stream(side[0]):
        knob(6); one()
    main.wait_stream(side[0])
    knob(7)


t = timed(two); two(); torch.cuda.synchronize()
print("two streams (light | rest)  %.3f ms   max |diff| %.2e" % (t, float((out - ref).abs().max())))
