#!/usr/bin/env python3
"""End-to-end DCASEEvaluator throughput on a synthetic validation set (10 s clips): device-side mixing + prefetch +
caption cache (default) vs the reference's host-side numpy mixing.  Usage (GPU box): python tools/eval_bench.py [N]"""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lass_amd import synthetic
from lass_amd.audiosep import AudioSep, PrecomputedQueryEncoder
from lass_amd.evaluator import DCASEEvaluator
from lass_amd.resunet import ResUNet30

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tmp = tempfile.mkdtemp()
csv_path = synthetic.write_validation_set(tmp, n_clips=n, length=160000)
sd = synthetic.make_state_dict()
m = ResUNet30(1, 1, 512)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
pl = AudioSep(ss_model=m.to("cuda:0").eval(), query_encoder=PrecomputedQueryEncoder())
res = {}
for name, kw in (("device_mixing", {}), ("host_mixing", {"device_mixing": False, "io_workers": 1})):
    ev = DCASEEvaluator(16000, csv_path, os.path.join(tmp, "lass_validation"), batch_size=16, **kw)
    ev(pl)  # warm-up (file cache, workspace)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = ev(pl)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[name] = (n / dt, out)
    print(f"{name:14s} {n / dt:8.1f} clips/s  (SISDR, SDRi, SDR) = {tuple(round(v, 3) for v in out)}", flush=True)
