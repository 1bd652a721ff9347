#!/usr/bin/env python3
"""Benchmark of the separation hot path (BASELINE.json metric: clips/sec, one clip = 10 s @ 16 kHz).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path - lass_separate: STFT -> FiLM ResUNet30 -> mask -> iSTFT - over one batch of
synthetic mixtures already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]: ResUNet30 fp32, batch 16,
10 s @ 16 kHz, fixed (precomputed) condition embeddings, seeded random-init weights.  For N>1 every rank separates its
own 16 clips (weak scaling; clip-level sharding, no data-path collective); value = clips of all ranks / max-over-ranks
time.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def host_cores() -> int:
    """CPU threads this process may actually use: affinity mask, capped by the cgroup CPU quota; a GPU box exposes all
    of the host's logical CPUs but grants a 16-core share per GPU, and over-subscribing it makes the baseline
    meaningless."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    env = os.environ.get("LASS_BENCH_CPU_THREADS")
    if env:
        return int(env)
    return min(n, 16)


def pmc_traffic():
    """HBM bytes per conv3x3 launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command (profiles/rNN/conv_traffic.json, made by tools/traffic_summary.py); bench.py itself cannot read PMCs."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "conv_traffic.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    return d.get("traffic_bytes_per_launch"), {"source": os.path.relpath(files[-1], ROOT),
                                               "raw_bytes_per_launch": d.get("traffic_bytes_per_launch_raw"),
                                               "note": d.get("note")}


def cpu_baseline(sd, length, seconds_budget=20.0):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores: a bounded sample of the same
    workload (batch-1 forwards of 10 s clips until ~seconds_budget of CPU time is spent)."""
    from lass_amd import synthetic
    from oracle import resunet as orr
    cores = host_cores()
    torch.set_num_threads(cores)
    osd = orr.to_torch(sd)
    _, mix = synthetic.make_mixtures(1, length)
    inp = {"mixture": torch.from_numpy(mix)[:, None, :], "condition": torch.from_numpy(synthetic.make_condition(1))}
    orr.forward(osd, inp)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        orr.forward(osd, inp)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_budget or n >= 50:
            break
    return {"value": n / dt, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{n} batch-1 forwards of one {length / 16000:.0f} s clip, oracle/resunet.py (torch-CPU fp32), "
                      f"{torch.get_num_threads()} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="clips per GPU per step (BASELINE configs[1]: 16)")
    ap.add_argument("--length", type=int, default=160000, help="samples per clip (10 s @ 16 kHz)")
    ap.add_argument("--dtype", choices=["f32", "bf16", "bf16x3"], default="f32",
                    help="f32 = BASELINE configs[1] (headline); bf16 = configs[2] (bf16-MFMA convolutions)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    import __graft_entry__ as ge
    ge.build()
    from lass_amd import arch, synthetic
    from lass_amd.resunet import ResUNet30

    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    sd = synthetic.make_state_dict()
    model = ResUNet30(1, 1, 512)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model = model.to(dev).eval().set_compute_dtype(args.dtype)
    eng = model.engine
    B, L = args.batch, args.length
    # distinct synthetic clips per rank; a small pool tiled to B keeps host-side generation short
    pool = min(B, 4)
    _, mix = synthetic.make_mixtures(pool, L, first=rank * pool)
    mix = np.concatenate([mix] * ((B + pool - 1) // pool))[:B]
    mixture = torch.from_numpy(mix).to(dev)
    cond = torch.from_numpy(synthetic.make_condition(B)).to(dev)
    out = torch.empty_like(mixture)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        eng.separate(mixture, cond, out)
    torch.cuda.synchronize()
    eng.set_profiling(True)   # HIP events around every kernel class, on the launch stream, inside the timed region
    eng.profile(reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.separate(mixture, cond, out)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    prof = eng.profile(reset=True)
    eng.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out).all()

    if rank == 0:
        rows = arch.conv_layer_table(arch.padded_frames(arch.frames_for(L)))
        conv_flops = 2.0 * B * sum(r["macs"] for r in rows if r["kind"] == "3x3" or r["name"].endswith(".shortcut"))
        total_flops = 2.0 * B * sum(r["macs"] for r in rows)
        ms, launches = prof["conv3x3_mfma"]
        per_step_ms = ms / args.steps
        achieved = conv_flops / (per_step_ms * 1e-3) / 1e12 if per_step_ms > 0 else 0.0
        traffic, traffic_meta = pmc_traffic() if (B, L) == (16, 160000) else (None, None)
        # With the Winograd kernels (default) the 3x3 convs (even H) execute 4 instead of 9 multiplies per output and
        # (cin, cout): `achieved` stays the ALGORITHMIC (direct-convolution) rate and may exceed the MFMA peak; the
        # rate the matrix pipe really sustains is `executed_tflops`.
        wino = os.environ.get("LASS_WINO", "1") != "0"
        exec_flops = 0.0
        for r in rows:
            if r["kind"] == "3x3":
                exec_flops += 2.0 * B * r["macs"] * ((4.0 / 9.0) if (wino and r["h"] % 2 == 0) else 1.0)
            elif r["name"].endswith(".shortcut"):  # transform-domain 1x1: 4 of 16 xi per 2x2 tile = direct-1x1 count
                exec_flops += 2.0 * B * r["macs"]
        executed = exec_flops / (per_step_ms * 1e-3) / 1e12 if per_step_ms > 0 else 0.0
        peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
        conv_bytes = float(B) * arch.conv3x3_bytes_per_clip(L)
        hbm_gbs = conv_bytes / (per_step_ms * 1e-3) / 1e9 if per_step_ms > 0 else 0.0
        res = {
            "metric": "clips/sec (10s@16kHz)", "value": world * B * args.steps / dt, "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"ResUNet30 separate() {'fp32' if args.dtype == 'f32' else 'bf16-MFMA convs (f32 storage/accumulate)'}, "
                                   f"batch={B}/GPU, {L / 16000:.0f}s@16kHz clips, fixed precomputed condition embedding, "
                                   f"seeded random-init weights (BASELINE configs[{1 if args.dtype == 'f32' else 2}])",
                       "clips_per_gpu_per_step": B, "samples_per_clip": L, "parallelism": f"clip-sharded x{world}"},
            "realtime_factor": world * B * args.steps / dt * (L / 16000.0),
            "roofline": {"bound": "mfma", "kernel": "conv3x3_mfma (3x3 convs + fused 1x1 shortcuts, f32 MFMA; achieved = algorithmic direct-conv FLOPs)",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic if args.dtype == "f32" else None,
                         "traffic_meta": traffic_meta,
                         "algorithm": ("bf16 MFMA direct conv, f32 activations converted while staged"
                                       + (" (hi+lo split operands, 3 MFMAs per product)" if args.dtype == "bf16x3" else ""))
                                      if args.dtype != "f32" else
                                      ("Winograd F(2x2,3x3) on the f32 MFMA (all 3x3 convs; 1x1 shortcuts in the transform domain)" if wino else "direct"),
                         "executed_tflops": executed if args.dtype == "f32" else achieved,
                         "executed_frac": (executed if args.dtype == "f32" else achieved) / peak,
                         "hbm_algorithmic_gbs": hbm_gbs, "hbm_peak_gbs": PEAK_HBM_GBS, "hbm_frac": hbm_gbs / PEAK_HBM_GBS,
                         "hbm_model": "compulsory bytes with f32 activation storage (SURVEY 8d); the bf16 mode keeps block "
                                      "intermediates, decoder concats and pooled outputs as bf16, so it moves fewer bytes",
                         "launches_per_step": launches / args.steps, "avg_launch_ms": ms / max(1, launches),
                         "algorithmic_gflop_per_step": conv_flops / 1e9,
                         "whole_step_tflops": total_flops / (dt / args.steps) / 1e12},
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items()},
        }
        if world == 1 and not args.no_cpu_baseline and args.dtype == "f32":
            res["cpu_baseline"] = cpu_baseline(sd, L, args.cpu_seconds)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
