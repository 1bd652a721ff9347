import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lass_amd import synthetic
from lass_amd.engine import Engine
B, H, W = 2, 40, 64
sd = synthetic.make_state_dict()
g = torch.Generator().manual_seed(1)
x1 = torch.randn(B, 32, H, W, generator=g); x6 = torch.randn(B, 64, H, W, generator=g)
cond = torch.from_numpy(synthetic.make_condition(B))
outs = {}
for sw in ("1", "0"):
    if sw == "0": os.environ.pop("LASS_W32_KINDS", None)
    os.environ["LASS_WINO32"] = sw
    e = Engine("cuda:0"); e.load_state_dict(sd)
    shift = e.film(cond.cuda())
    y1, p1 = e.encoder_block("base.encoder_block1", x1.cuda(), shift, 32, (2, 2))
    y6 = e.convblock("base.decoder_block6.conv_block2", x6.cuda(), shift, 32)
    outs[sw] = (y1.cpu(), p1.cpu(), y6.cpu())
for name, a, b in zip(("enc1", "pool", "dec6"), outs["1"], outs["0"]):
    d = (a - b).abs()
    print(name, "max diff", float(d.max()), "ref rms", float(b.pow(2).mean().sqrt()))
    print("  per batch", d.amax(dim=(1, 2, 3)).tolist())
    print("  per channel (first 8)", [round(v, 4) for v in d.amax(dim=(0, 2, 3)).tolist()[:8]])
    print("  rows with error", [i for i, v in enumerate(d.amax(dim=(0, 1, 3)).tolist()) if v > 1e-4])
    print("  cols with error", [i for i, v in enumerate(d.amax(dim=(0, 1, 2)).tolist()) if v > 1e-4])
