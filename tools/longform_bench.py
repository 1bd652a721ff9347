import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from lass_amd import synthetic
from lass_amd.resunet import ResUNet30
sd = synthetic.make_state_dict()
m = ResUNet30(1, 1, 512); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.to('cuda:0').eval()
L = 960000
x = torch.from_numpy(np.random.default_rng(0).standard_normal(L).astype(np.float32) * 0.1)[None, None].cuda()
c = torch.from_numpy(synthetic.make_condition(1)).cuda()
for mb in (1, 16):
    m.chunk_inference({'mixture': x, 'condition': c}, max_batch=mb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): y = m.chunk_inference({'mixture': x, 'condition': c}, max_batch=mb)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f'long-form 30 s @ 32 kHz chunk_inference max_batch={mb}: {dt*1e3:.1f} ms  ({30.0/dt:.0f}x real time)', flush=True)
