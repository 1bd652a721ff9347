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
cases = [("device_mixing", {}), ("host_mixing", {"device_mixing": False, "io_workers": 1})]
if os.environ.get("EVAL_BENCH_SWEEP"):  # decode-thread sweep
    cases = [(f"device_mixing_w{w}", {"io_workers": w}) for w in (2, 4, 8)] + [(f"host_mixing_w{w}", {"device_mixing": False, "io_workers": w}) for w in (1,)]
for name, kw in cases:
    ev = DCASEEvaluator(16000, csv_path, os.path.join(tmp, "lass_validation"), batch_size=16, **kw)
    ev(pl)  # warm-up (file cache, workspace)
    dts = []
    for _ in range(int(os.environ.get("EVAL_BENCH_REPS", "3"))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ev(pl)
        torch.cuda.synchronize(); dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[len(dts) // 2]  # median call
    res[name] = (n / dt, out)
    print(f"{name:14s} {n / dt:8.1f} clips/s (median of {len(dts)} calls: {' '.join(f'{n / d:.0f}' for d in dts)})  (SISDR, SDRi, SDR) = {tuple(round(v, 3) for v in out)}", flush=True)

# the separator alone on a resident batch, same process and box: the yardstick for the evaluator's rate
_, mix = synthetic.make_mixtures(16, 160000)
inp = {"mixture": torch.from_numpy(mix)[:, None, :].cuda(), "condition": torch.from_numpy(synthetic.make_condition(16)).cuda()}
with torch.no_grad():
    for _ in range(5): pl.ss_model(inp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): pl.ss_model(inp)
    torch.cuda.synchronize(); sep = 16 * 20 / (time.perf_counter() - t0)
print(f"separator alone {sep:8.1f} clips/s  -> evaluator / separator = " + ", ".join(f"{k} {v[0] / sep:.3f}" for k, v in res.items()), flush=True)
