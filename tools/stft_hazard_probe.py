#!/usr/bin/env python3
"""DESIGN.md section 5b, the open question: when an STFT launch comes out wrong beside a bf16 lass_separate, is the first
wrong value one the butterflies READ from LDS (the LDS path of the victim) or one they COMPUTED from right inputs (the packed
vector ALU)?

Needs the diagnostic library: stft.hip built with -DLASS_STFT_DBG and WITHOUT -fno-slp-vectorize (so that its butterflies are
v_pk_*_f32, the build that fails), linked with the shipped objects:
    LASS_HIP_LIB=lass_amd/csrc/liblass_hip_stftdbg.so python tools/stft_hazard_probe.py [K launches per trial] [trials]
Every radix-4 pass of stft2_kernel records, per workgroup, the four points and three twiddles as read from LDS and the four
results as computed (in front of their LDS stores).  The same launch run ALONE gives the reference records; for every launch
that ran beside the separation and whose spectrum differs, the FIRST differing record of each wrong workgroup is classified:
    read   : a point read in pass p differs although every result computed in pass p-1 was right  -> LDS write / read path
    twiddle: a twiddle read from the LDS table differs                                           -> LDS read path
    compute: a result of pass p differs although its four points and three twiddles were right    -> vector ALU
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lass_amd import _lib, arch, synthetic  # noqa: E402
from lass_amd.resunet import ResUNet30  # noqa: E402

N, NPASS = 1024, 5


def main(K=10, trials=4):
    lib = _lib.load()
    setbuf = lib.lass_dbg_stft_buffer
    setbuf.restype = ctypes.c_int
    setbuf.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    B, L = 8, 160000
    _, mix = synthetic.make_mixtures(4, L)
    mix = np.concatenate([mix] * 4)
    xa = torch.from_numpy(mix[:B]).cuda()
    xb = torch.from_numpy(mix[B:2 * B] * 0.7 + 0.01).cuda()
    cond = torch.from_numpy(synthetic.make_condition(B)).cuda()
    m = ResUNet30(1, 1, 512)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict().items()})
    os.environ["LASS_SPLIT"] = "0"
    e = m.cuda().eval().set_compute_dtype("bf16").engine
    e.set_graph_replay(False)
    T = arch.frames_for(L)
    nwg = ((arch.padded_frames(T) + 1) // 2) * B            # grid of stft2_kernel: (Tpad + 1) / 2 frame pairs x B clips
    rec_per_wg = NPASS * 3 * N                              # float2 records per workgroup
    mk = lambda: torch.zeros(nwg * rec_per_wg, 2, dtype=torch.float32, device="cuda")  # noqa: E731
    ref_rec = mk()
    stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    assert setbuf(ref_rec.data_ptr(), stream()) == 0
    ref = [t.clone() for t in e.front_end(xb)]
    assert setbuf(None, stream()) == 0
    oa = e.separate(xa, cond).clone()
    torch.cuda.synchronize()
    recs = [mk() for _ in range(K)]
    outs = [e.front_end(xb) for _ in range(K)]             # output sets allocated once, outside the overlapped region
    torch.cuda.synchronize()
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    tally = {"read": 0, "twiddle": 0, "compute": 0}
    by_pass = {}
    wrong_launches = 0
    detail_printed = 0
    unrecorded = 0
    read0_printed = 0
    lanes, comps, quarters, nslots = {}, {"x only": 0, "y only": 0, "both": 0}, {}, {}
    for trial in range(trials):
        for r in recs:
            r.zero_()
        torch.cuda.synchronize()
        endA = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sA):
            for _ in range(6):                              # keep stream A busy for the whole window
                e.separate(xa, cond)
            endA.record()
        done = []
        with torch.cuda.stream(sB):
            for i in range(K):
                assert setbuf(recs[i].data_ptr(), stream()) == 0
                e.front_end(xb, out=outs[i])
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                done.append(ev)
        torch.cuda.synchronize()
        print(f"trial {trial}: {sum(1 for ev in done if ev.elapsed_time(endA) > 0)} of {K} launches ended before stream A's separations did", flush=True)
        bad = [i for i, o in enumerate(outs) if not all(torch.equal(a, b) for a, b in zip(o, ref))]
        wrong_launches += len(bad)
        print(f"trial {trial}: {len(bad)} of {K} front-end launches wrong {bad}", flush=True)
        for i in bad:
            d = (recs[i] != ref_rec).any(dim=1).view(nwg, NPASS, 3, N)      # (wg, pass, kind, slot)
            wgs = torch.nonzero(d.flatten(1).any(dim=1))[:, 0].tolist()
            got5 = recs[i].view(nwg, NPASS, 3, N, 2)
            ref5 = ref_rec.view(nwg, NPASS, 3, N, 2)
            for wg in wgs:
                dw = d[wg]                                                   # (pass, kind, slot)
                if int(dw[0, 0].sum()) > N // 2:
                    # (nearly) every point this workgroup read in pass 0 differs: these are the records of ANOTHER INPUT - stream
                    # A's lass_separate runs the same stft2_kernel on its own mixtures and records into whatever buffer is set at
                    # that moment.  A harness artefact, not a wrong value: left out of the tally.
                    unrecorded += 1
                    continue
                first = None
                for ps in range(NPASS):
                    for kind in (0, 1, 2):
                        if bool(dw[ps, kind].any()):
                            first = (ps, kind)
                            break
                    if first:
                        break
                ps, kind = first
                cls = ("read", "twiddle", "compute")[kind]
                tally[cls] += 1
                by_pass[(ps, cls)] = by_pass.get((ps, cls), 0) + 1
                slots_t = torch.nonzero(dw[ps, kind])[:, 0]
                for s_ in slots_t.tolist():
                    lanes[s_ % 64] = lanes.get(s_ % 64, 0) + 1
                    quarters[s_ // 256] = quarters.get(s_ // 256, 0) + 1
                nslots[int(slots_t.numel())] = nslots.get(int(slots_t.numel()), 0) + 1
                dx = (got5[wg, ps, kind, slots_t, 0] != ref5[wg, ps, kind, slots_t, 0])
                dy = (got5[wg, ps, kind, slots_t, 1] != ref5[wg, ps, kind, slots_t, 1])
                comps["x only"] += int((dx & ~dy).sum()); comps["y only"] += int((~dx & dy).sum()); comps["both"] += int((dx & dy).sum())
                if kind == 0 and ps == 0 and read0_printed < 3:
                    read0_printed += 1
                    sl = slots_t[:6].tolist()
                    print(f"   [pass-0 read] launch {i} wg {wg} ({int(slots_t.numel())} slots):",
                          [(s_, [round(v, 6) for v in got5[wg, 0, 0, s_].tolist()], [round(v, 6) for v in ref5[wg, 0, 0, s_].tolist()]) for s_ in sl],
                          "| this launch's spectra differ from the reference in", int((outs[i][0] != ref[0]).sum()), "of", ref[0].numel(), "magnitudes", flush=True)
                if detail_printed < 6:
                    detail_printed += 1
                    slots = torch.nonzero(dw[ps, kind])[:, 0].tolist()
                    thr = sorted({s_ % 256 for s_ in slots})
                    print(f"   launch {i} wg {wg}: first wrong record = pass {ps} kind {cls}; {len(slots)} slots, threads "
                          f"{thr[:16]}{'...' if len(thr) > 16 else ''} (waves {sorted({t // 64 for t in thr})}), quarter(s) "
                          f"{sorted({s_ // 256 for s_ in slots})}", flush=True)
                    s0 = slots[0]
                    a = recs[i].view(nwg, NPASS, 3, N, 2)[wg, ps, kind, s0].tolist()
                    b = ref_rec.view(nwg, NPASS, 3, N, 2)[wg, ps, kind, s0].tolist()
                    print(f"      slot {s0}: got {a} expected {b}", flush=True)
                    if kind == 0 and ps > 0:
                        prev_ok = not bool(dw[ps - 1, 2].any())
                        print(f"      results computed in pass {ps - 1} all right: {prev_ok}", flush=True)
    print("workgroup records overwritten by stream A's own STFT launch (left out):", unrecorded)
    print("of the first wrong records: lanes (of 64) ->", dict(sorted(lanes.items())), "| result slot (0..3 = r0..r3 of the butterfly) ->",
          dict(sorted(quarters.items())), "| component ->", comps, "| wrong slots per workgroup ->", dict(sorted(nslots.items())), flush=True)
    print("wrong launches:", wrong_launches, "| first wrong record per wrong workgroup:", tally, "| by (pass, class):",
          dict(sorted(by_pass.items())), flush=True)
    return tally


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 4)
