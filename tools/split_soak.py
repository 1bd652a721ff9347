#!/usr/bin/env python3
"""Half-batch overlap (LASS_SPLIT) against the unsplit run: bit-for-bit, call after call.
usage: python tools/split_soak.py [mode] [calls]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lass_amd import synthetic  # noqa: E402
from lass_amd.resunet import ResUNet30  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B, L = 16, 160000
_, mix = synthetic.make_mixtures(B, L)
x = torch.from_numpy(mix).cuda()
cond = torch.from_numpy(synthetic.make_condition(B)).cuda()
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict().items()}


def make(split):
    os.environ["LASS_SPLIT"] = "2" if split == "1" else "0"  # 2: eager launches split too (the default splits in graphs only)
    m = ResUNet30(1, 1, 512)
    m.load_state_dict(sd)
    return m.cuda().eval().set_compute_dtype(mode).engine


e1, e0 = make("1"), make("0")
ref = e0.separate(x, cond).clone()
bad = 0
for graph in (False, True):
    e1.set_graph_replay(graph)
    for i in range(calls):
        o = e1.separate(x, cond)
        torch.cuda.synchronize()
        if not torch.equal(o, ref):
            bad += 1
            print(mode, "graph" if graph else "eager", "call", i, "differs: clips", torch.nonzero((o != ref).any(1))[:, 0].tolist(), flush=True)
out = torch.empty_like(ref)
for graph in (True, False):
    for e, name in ((e0, "unsplit"), (e1, "split")):
        e.set_graph_replay(graph)
        for _ in range(5): e.separate(x, cond, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): e.separate(x, cond, out=out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(mode, name, "graph replay" if graph else "eager launches", "%.3f ms/step  %.0f clips/s" % (dt / 20 * 1e3, B * 20 / dt), flush=True)
print(mode, "calls differing from the unsplit run:", bad, "of", 2 * calls, flush=True)
sys.exit(1 if bad else 0)
