#!/usr/bin/env python3
"""Long-form throughput (BASELINE configs[4]: 30 s @ 32 kHz clips, L = 960 000) on one MI355X:
  * ResUNet30.chunk_inference (resunet.py:655-714) window by window and 16 windows per launch;
  * ResUNet30 whole-clip forward;
  * the multi-STFT separator (lass_amd/resunet_with_multistft.py) whole-clip forward, B = 1 and B = 2.
Prints one line per case: ms per clip, clips/s, x real time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lass_amd import arch, synthetic
from lass_amd.resunet import ResUNet30
from lass_amd.resunet_with_multistft import ResUNet30 as MultiSTFT

L = 960000
x = torch.from_numpy(np.random.default_rng(0).standard_normal(L).astype(np.float32) * 0.1)[None, None].cuda()


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


sd = synthetic.make_state_dict()
m = ResUNet30(1, 1, 512); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.to('cuda:0').eval()
c = torch.from_numpy(synthetic.make_condition(1)).cuda()
for mb in (1, 16):
    dt = timeit(lambda: m.chunk_inference({'mixture': x, 'condition': c}, max_batch=mb))
    print(f'ResUNet30 chunk_inference max_batch={mb}: {dt*1e3:.1f} ms/clip  {1/dt:.1f} clips/s  ({30.0/dt:.0f}x real time)', flush=True)
dt = timeit(lambda: m({'mixture': x, 'condition': c}))
print(f'ResUNet30 whole-clip forward B=1: {dt*1e3:.1f} ms/clip  {1/dt:.1f} clips/s  ({30.0/dt:.0f}x real time)', flush=True)
del m
torch.cuda.empty_cache()
ms = MultiSTFT(1, 1, 512)
ms.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict_ms().items()})
ms = ms.to('cuda:0').eval()
gmac = sum(r['macs'] for r in arch.ms_conv_layer_table(arch.padded_frames(arch.frames_for(L)))) / 1e9
for B in (1, 2):
    xb = x.expand(B, -1, -1).contiguous()
    cb = torch.from_numpy(synthetic.make_condition(B)).cuda()
    dt = timeit(lambda: ms({'mixture': xb, 'condition': cb}))
    print(f'multi-STFT separator whole-clip forward B={B}: {dt/B*1e3:.1f} ms/clip  {B/dt:.2f} clips/s  ({30.0*B/dt:.0f}x real time)  '
          f'{2*gmac*B/dt/1e3:.1f} algorithmic TFLOP/s ({gmac:.0f} GMAC/clip), workspace {ms.engine.workspace_bytes(B, L)/2**30:.1f} GiB', flush=True)
# the same model and clip in the bf16 compute modes (f32 tensors between the blocks)
c1 = torch.from_numpy(synthetic.make_condition(1)).cuda()
for mode in ("bf16x3", "bf16"):
    ms.set_compute_dtype(mode)
    dt = timeit(lambda: ms({'mixture': x, 'condition': c1}), n=5)
    print(f'multi-STFT separator whole-clip forward B=1, compute_dtype={mode}: {dt*1e3:.1f} ms/clip  {1/dt:.1f} clips/s  ({30.0/dt:.0f}x real time)', flush=True)
