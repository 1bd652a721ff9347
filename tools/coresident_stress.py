#!/usr/bin/env python3
"""Reproducer / regression check for the gfx950 co-residency hazard of DESIGN.md section 5b.

Stream A runs lass_separate (B = 8) in the given compute mode; stream B meanwhile launches the STFT front end NREP times, each
into its own buffers.  Every front-end result must equal the one computed alone.  With stft.hip built WITH packed-f32
instructions (hipcc's default: the SLP vectoriser turns the complex butterflies into v_pk_{mul,fma,add}_f32) about one launch
in six came out wrong in bf16 mode - a few hundred of the 8008 frames each, always a run of 4-16 neighbouring points of one
FFT pass - whenever a bf16 conv kernel (v_mfma_f32_32x32x16_bf16 fed from LDS by ds_read_b128) was resident on the same CU;
never in f32 mode, never beside torch's own kernels, never with the MFMA or the LDS reads alone.  Built as __graft_entry__.py
builds it (-fno-slp-vectorize for the kernels without matrix instructions) every launch is exact.

usage: python tools/coresident_stress.py [bf16|f32|bf16x3] [NREP]     (LASS_HIP_LIB=<other build> to test a variant)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lass_amd import synthetic  # noqa: E402
from lass_amd.resunet import ResUNet30  # noqa: E402


def run(mode="bf16", nrep=120, trials=3, verbose=True, nsep=8, return_overlap=False):
    B, L = 8, 160000
    _, mix = synthetic.make_mixtures(4, L)
    mix = np.concatenate([mix] * 4)
    xa = torch.from_numpy(mix[:B]).cuda()
    xb = torch.from_numpy(mix[B:2 * B] * 0.7 + 0.01).cuda()
    cond = torch.from_numpy(synthetic.make_condition(B)).cuda()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict().items()}
    m = ResUNet30(1, 1, 512)
    m.load_state_dict(sd)
    saved = os.environ.get("LASS_SPLIT")
    os.environ["LASS_SPLIT"] = "0"  # read when the context is created: stream A runs ONE ordinary unsplit separation
    try:
        e = m.cuda().eval().set_compute_dtype(mode).engine
    finally:
        if saved is None:
            del os.environ["LASS_SPLIT"]
        else:
            os.environ["LASS_SPLIT"] = saved
    e.set_graph_replay(False)
    ref = [t.clone() for t in e.front_end(xb)]
    oa = e.separate(xa, cond).clone()
    # Output sets are allocated ONCE, outside the overlapped region (a front_end that allocates its four 16-MB tensors paces the
    # launches with hipMalloc and may never run beside a conv kernel); stream A is kept busy with `nsep` separations for the whole
    # window, and events on both streams say how many front-end launches really ended while stream A was still separating.
    outs = [e.front_end(xb) for _ in range(nrep)]
    torch.cuda.synchronize()
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    wrong, overlapped = 0, 0
    for trial in range(trials):
        for o in outs:
            for t in o:
                t.zero_()
        torch.cuda.synchronize()
        startA, endA = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        done = []
        with torch.cuda.stream(sA):
            startA.record()
            ra = None
            for _ in range(nsep):
                ra = e.separate(xa, cond)
            endA.record()
        with torch.cuda.stream(sB):
            for i in range(nrep):
                e.front_end(xb, out=outs[i])
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                done.append(ev)
        torch.cuda.synchronize()
        beside = sum(1 for ev in done if startA.elapsed_time(ev) > 0 and ev.elapsed_time(endA) > 0)
        overlapped += beside
        bad = [i for i, o in enumerate(outs) if not all(torch.equal(a, b) for a, b in zip(o, ref))]
        wrong += len(bad) + (0 if torch.equal(ra, oa) else 1)
        if verbose:
            print(mode, "trial", trial, "front-end launches with a wrong result:", len(bad), "of", nrep, bad[:24],
                  "| ended while stream A was separating:", beside, "| lass_separate beside them exact:", torch.equal(ra, oa), flush=True)
            if bad:
                d = outs[bad[0]][0] != ref[0]  # (B, T, 513) magnitudes
                per_frame = d.sum(-1).flatten()
                fr = torch.nonzero(per_frame)[:, 0]
                print("   launch", bad[0], ": frames affected", int(fr.numel()), "of", per_frame.numel(), "| bins differing per frame: min",
                      int(per_frame[fr].min()), "median", int(per_frame[fr].median()), "max", int(per_frame[fr].max()))
    if return_overlap:
        return wrong, overlapped
    return wrong


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    sys.exit(1 if run(mode, nrep) else 0)
