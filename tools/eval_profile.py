#!/usr/bin/env python3
"""Where the evaluator's wall time goes (host side): cProfile of one DCASEEvaluator call over N synthetic 10 s clips.
Usage (GPU box): python tools/eval_profile.py [N]"""
import cProfile, os, pstats, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lass_amd import synthetic
from lass_amd.audiosep import AudioSep, PrecomputedQueryEncoder
from lass_amd.evaluator import DCASEEvaluator
from lass_amd.resunet import ResUNet30

n = int(sys.argv[1]) if len(sys.argv) > 1 else 260
tmp = tempfile.mkdtemp()
csv_path = synthetic.write_validation_set(tmp, n_clips=n, length=160000)
sd = synthetic.make_state_dict()
m = ResUNet30(1, 1, 512)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
pl = AudioSep(ss_model=m.to("cuda:0").eval(), query_encoder=PrecomputedQueryEncoder())
ev = DCASEEvaluator(16000, csv_path, os.path.join(tmp, "lass_validation"), batch_size=16)
ev(pl); ev(pl)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable(); ev(pl); pr.disable()
torch.cuda.synchronize()
print(f"{n / (time.perf_counter() - t0):.1f} clips/s under cProfile")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
