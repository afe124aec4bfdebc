#!/usr/bin/env python3
"""Wall time of whole training epochs of the drop-in script on a ZINC-250k-sized synthetic set (train 239455 + val
10000 molecules, train_inductive.py:79): python tools/r02/epoch_time.py [batch sizes...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gae_dgl_amd import train_inductive as TI

for b in (int(x) for x in (sys.argv[1:] or ["128", "4096"])):
    for capture in ("auto", "off"):
        t0 = time.perf_counter()
        marks = []
        orig = TI._run_epoch

        def timed(*a, **k):
            torch.cuda.synchronize(); t = time.perf_counter()
            out = orig(*a, **k)
            torch.cuda.synchronize(); marks.append(time.perf_counter() - t)
            return out
        TI._run_epoch = timed
        try:
            TI.main(["--hidden_dims", "32", "16", "--synthetic", "249455", "-b", str(b), "-e", "3", "--seed", "0",
                     "-s", "/tmp/epoch_time", "--no_plot", "--capture", capture])
        finally:
            TI._run_epoch = orig
        tr, va = marks[0::2], marks[1::2]
        print(f"batch {b:5d} capture={capture:4s}: train epochs {['%.3f' % t for t in tr]} s, validation passes "
              f"{['%.3f' % t for t in va]} s, whole run {time.perf_counter() - t0:.1f} s (incl. synthetic data + upload)",
              flush=True)
