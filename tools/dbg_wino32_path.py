"""Whole-path check of wino32.hip, one launch kind at a time: the separator on a tiny mixture with LASS_WINO32=0 (wino.hip) as
the reference, then with only the kinds of LASS_W32_KINDS routed to wino32.  Each configuration runs in its own process
(the switches are read once).  usage: python tools/dbg_wino32_path.py [L]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from lass_amd import synthetic
    from lass_amd.resunet import ResUNet30
    L = int(sys.argv[3])
    sd = synthetic.make_state_dict()
    m = ResUNet30(1, 1, 512)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.to("cuda:0").eval()
    _, mix = synthetic.make_mixtures(2, L)
    cond = synthetic.make_condition(2)
    out = m({"mixture": torch.from_numpy(mix)[:, None, :].cuda(), "condition": torch.from_numpy(cond).cuda()})["waveform"]
    np.save(sys.argv[2], out.cpu().numpy())
    sys.exit(0)
import numpy as np
L = sys.argv[1] if len(sys.argv) > 1 else "16000"
def run(tag, **env):
    e = dict(os.environ); e.update(env)
    path = f"/tmp/dbg_w32_{tag}.npy"
    subprocess.run([sys.executable, __file__, "--child", path, L], check=True, env=e)
    return np.load(path)
ref = run("ref", LASS_WINO32="0")
names = {0: "CONV1_ACT (dec6.conv1)", 4: "CONV1_ACT_PRE (enc1.conv1)", 5: "CONV2_IDENT_PRE (enc1.conv2)", 2: "CONV2_SHORTCUT (dec6.conv2+mask)"}
for k, n in names.items():
    out = run(f"k{k}", LASS_WINO32="1", LASS_W32_KINDS=str(1 << k))
    print(f"kind {k} {n}: rms diff {np.sqrt(np.mean((out - ref) ** 2)):.3e}  (signal rms {np.sqrt(np.mean(ref ** 2)):.3e})", flush=True)
