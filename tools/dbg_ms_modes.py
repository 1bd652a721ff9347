"""The multi-STFT separator in the bf16 / bf16x3 compute modes against its f32 output and the oracle (GPU box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lass_amd import synthetic
from lass_amd.resunet_with_multistft import ResUNet30
from oracle import resunet as orr, resunet_multistft as oms
sd = synthetic.make_state_dict_ms()
m = ResUNet30(1, 1, 512)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
m = m.to("cuda:0").eval()
rms = lambda a: float(a.double().pow(2).mean().sqrt())
for B, L in ((2, 16000), (1, 8077), (1, 64000), (1, 960000)):
    _, mix = synthetic.make_mixtures(B, L)
    cond = synthetic.make_condition(B)
    inp = {"mixture": torch.from_numpy(mix)[:, None].cuda(), "condition": torch.from_numpy(cond).cuda()}
    ref = oms.forward(orr.to_torch(sd), {"mixture": torch.from_numpy(mix)[:, None], "condition": torch.from_numpy(cond)})["waveform"] if L <= 16000 else None
    outs = {}
    for mode in ("f32", "bf16x3", "bf16"):
        m.set_compute_dtype(mode)
        try:
            outs[mode] = m(inp)["waveform"].cpu()
        except Exception as e:
            print(B, L, mode, "ERROR", e); continue
        msg = f"B={B} L={L} {mode}: rms vs f32 {rms(outs[mode] - outs['f32']):.3e} (signal {rms(outs['f32']):.3e})"
        if ref is not None: msg += f"  vs oracle {rms(outs[mode] - ref):.3e}"
        print(msg, flush=True)
