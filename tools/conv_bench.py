#!/usr/bin/env python3
"""Per-layer timing of the residual blocks / transposed convs at the BASELINE configs[1] shapes (B=16, T=1024).
Usage (GPU box): LASS_CONV_VARIANT=n python tools/conv_bench.py [--iters 5] [--only enc1,dec6]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lass_amd import arch, synthetic
from lass_amd.engine import Engine

ap = argparse.ArgumentParser(); ap.add_argument('--iters', type=int, default=5); ap.add_argument('--only', default='')
ap.add_argument('--batch', type=int, default=16)
a = ap.parse_args()
eng = Engine('cuda:0'); eng.load_state_dict(synthetic.make_state_dict(), os.environ.get('LASS_COMPUTE', 'f32'))
B = a.batch
shift = eng.film(torch.from_numpy(synthetic.make_condition(B)).cuda())
h, w = 1024, 512
jobs = []
sizes = []
for e in arch.ENCODERS:
    jobs.append((e.name, 'block', f'base.{e.name}.conv_block1', e.cin, e.cout, h, w, None))
    sizes.append((h, w)); h //= e.down[0]; w //= e.down[1]
sizes.pop()
for d in arch.DECODERS:
    jobs.append((d.name + '.up', 'up', f'base.{d.name}', d.cin, d.cout, h, w, d.up))
    h *= d.up[0]; w *= d.up[1]
    jobs.append((d.name, 'block', f'base.{d.name}.conv_block2', 2 * d.cout, d.cout, h, w, None))
only = [s for s in a.only.split(',') if s]
tot_ms = tot_fl = 0.0
res = {}
for name, kind, prefix, cin, cout, H, W, up in jobs:
    if only and not any(o in name for o in only): continue
    x = torch.randn(B, cin, H, W, device='cuda')
    if kind == 'block':
        macs = H * W * (9 * cin * cout + 9 * cout * cout + (cin * cout if cin != cout else 0))
        fn = lambda: eng.convblock(prefix, x, shift, cout)
    else:
        macs = H * W * cin * cout * up[0] * up[1]
        fn = lambda: eng.upconv(prefix, x, shift, cout, up)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(a.iters):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts)); tf = 2 * B * macs / (ms * 1e-3) / 1e12
    tot_ms += ms; tot_fl += 2 * B * macs
    res[name] = (ms, tf)
    print(f'{name:22s} cin={cin:4d} cout={cout:4d} {H:5d}x{W:<4d} {ms:8.3f} ms {tf:7.1f} TF', flush=True)
    del x
print(f'TOTAL variant={os.environ.get("LASS_CONV_VARIANT","default")} {tot_ms:.3f} ms  {tot_fl/tot_ms/1e9:.1f} TF', flush=True)
