#!/usr/bin/env python3
"""Benchmark of the separation hot path (BASELINE.json metric: clips/sec, one clip = 10 s @ 16 kHz).

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 without WORLD_SIZE in the environment: this process touches no GPU, builds the
library in a child, starts N fresh ranks (`python -m torch.distributed.run --nproc-per-node N bench.py ...`, one rank
per GPU, backend "nccl" = RCCL), forwards rank 0's JSON line and exits with the launcher's code.  Launched under
torch.distributed.run directly (WORLD_SIZE set) it is one of those ranks.

A "step" is one pass of the hot path - lass_separate: STFT -> FiLM ResUNet30 -> mask -> iSTFT - over one batch of
synthetic mixtures already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]: ResUNet30 fp32, batch 16,
10 s @ 16 kHz, fixed (precomputed) condition embeddings, seeded random-init weights.  For N>1 every rank separates its
own 16 clips (weak scaling; clip-level sharding, no data-path collective: configs[3] at N=8); after the timed loop each
rank reduces its clips to SDR / SDRi / SI-SDR rows on the device and ONE all-gather of those rows ends the
job.  value = clips of all ranks / max-over-ranks time.  Prints ONE JSON line on rank 0.

Timing protocol: `value` comes from an un-instrumented loop; the per-kernel-class milliseconds (roofline) come from a
second, shorter loop with HIP events around every class on the launch stream (lass_set_profiling).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, 64 FLOP/clk/SIMD @ 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
# Winograd multiply counts per output pixel and (cin, cout) pair: F(2x2,3x3) 4 (16 per 2x2 tile), F(4x4,3x3) 2.25 (36 per 4x4
# tile), direct 9.  Which 3x3 layers run as F(4x4,3x3) is lass_amd.arch.wino4_routed (mirror of the C dispatch).


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["resunet30", "multistft"], default="resunet30",
                    help="resunet30 = BASELINE configs[1..3] (10 s @ 16 kHz clips, 16 per GPU); multistft = configs[4]'s "
                         "per-GPU job: the multi-STFT ResUNet on 30 s @ 32 kHz clips (defaults --batch 1 --length 960000), "
                         "the same timing / exchange protocol, so the driver's N=1,2,4,8 command scales it too")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU per step (default 16: BASELINE configs[1]; "
                                                            "1 with --workload multistft)")
    ap.add_argument("--length", type=int, default=None, help="samples per clip (default 160000 = 10 s @ 16 kHz; "
                                                             "960000 = 30 s @ 32 kHz with --workload multistft)")
    ap.add_argument("--dtype", choices=["f32", "bf16", "bf16x3"], default="f32",
                    help="headline arithmetic: f32 = BASELINE configs[1]; bf16 = configs[2] (bf16-MFMA convolutions)")
    ap.add_argument("--modes", default="auto",
                    help="extra compute modes timed after the headline region and reported under `modes` "
                         "(comma list of bf16,bf16x3; 'auto' = both at N=1 with --dtype f32, 'none' = skip)")
    ap.add_argument("--profile-steps", type=int, default=5, help="steps of the second, HIP-event-instrumented loop")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the one exchange step; gloo also lets ranks share a GPU "
                         "(rehearsal on a 1-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-short", action="store_true",
                    help="cpu_baseline at B=16 with 1 warm-up + 3 timed forwards instead of SURVEY 8(d)'s 3 + 10 (~3 min of CPU time)")
    ap.add_argument("--cpu-full", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--print-launch", action="store_true", help="N>1 parent: print the launch command and exit")
    args = ap.parse_args(argv)
    ms = args.workload == "multistft"
    if args.batch is None:
        args.batch = 1 if ms else 16
    if args.length is None:
        args.length = 960000 if ms else 160000
    return args


# ---- N > 1 parent: never touches the GPU ------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(argv, nproc: int, port: int):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)


def half_batch_overlap(mode: str, B: int) -> bool:
    """Whether lass_separate runs this batch as two overlapping half-batches (api.hip split_halves; include/lass_hip.h)."""
    return bool(os.environ.get("LASS_SPLIT", "1") != "0" and B >= 8 and B % 2 == 0)


def parent_launch(args, argv) -> int:
    """Spawn the ranks as fresh children (a GPU-initialised process must never exec or fork into ranks)."""
    cmd = launch_command(argv, args.gpus, free_port())
    if args.print_launch:
        print(json.dumps({"launch": cmd}))
        return 0
    # build once, in a child, so the ranks find an up-to-date library (they re-check under a file lock)
    subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, check=True,
                   stdout=sys.stderr)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    # stdout and stderr of the ranks in one stream: rank 0's JSON line is picked out, everything else goes to our stderr (on a
    # failure only its tail, marked, so that the failing rank's traceback is the last thing in the log)
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    line, rest = None, []
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            rest.append(ln)
    if p.returncode != 0:
        print(f"bench.py: the launcher exited with code {p.returncode}; last output of the ranks:", file=sys.stderr)
        rest = rest[-60:]
    for ln in rest:
        print(ln, file=sys.stderr)
    if p.returncode == 0 and line is None:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 1
    if line is not None and p.returncode == 0:
        print(line, flush=True)
    return p.returncode


# ---- helpers of a rank ---------------------------------------------------------------------------------------------------
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


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def pmc_traffic(dtype: str):
    """HBM bytes per launch of the dominant kernel class from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    passes of this same command (profiles/rNN/conv_traffic_<dtype>.json, made by tools/traffic_summary.py with the
    FETCH_SIZE calibration of tools/fetch_calib): bench.py itself cannot read PMCs.  The file carries the hash of the
    kernel sources it was measured at (`source_hash` = __graft_entry__._src_hash()); when that differs from the library
    in use the measurement says nothing about these kernels: `stale` is set and the caller reports traffic = null."""
    import glob
    import __graft_entry__ as ge
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"conv_traffic_{dtype}.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    d["source"] = "committed profile " + os.path.relpath(files[-1], ROOT)
    d["stale"] = d.get("source_hash") != ge._src_hash()
    return d


def traffic_fields(dtype, alg_bytes_launch, enabled=True):
    """`traffic*` entries of a roofline record: the PMC measurement when it was taken at the kernels in use, else null +
    `traffic_stale` (a committed number of OTHER kernels is not reported as this run's)."""
    t = pmc_traffic(dtype) if enabled else None
    if t is None:
        return {"traffic": None, "traffic_over_algorithmic": None, "traffic_source": None}
    if t["stale"]:
        return {"traffic": None, "traffic_over_algorithmic": None, "traffic_stale": True,
                "traffic_source": t["source"] + " (taken at other kernel sources: not reported)",
                "traffic_source_hash": t.get("source_hash")}
    return {"traffic": t["traffic_bytes_per_launch"],
            "traffic_over_algorithmic": t["traffic_bytes_per_launch"] / alg_bytes_launch,
            "traffic_stale": False, "traffic_source": t["source"], "traffic_source_hash": t["source_hash"],
            "traffic_fetch_calibration": t.get("fetch_calibration")}


def cpu_baseline(sd, length, full=True):
    """The oracle (CPU restatement of the reference path, kind "port") timed on this box's host cores, SURVEY 8(d)
    protocol: B=1 and B=16, each with 3 warm-up + 10 timed forwards, median (a B=16 forward takes ~14 s of CPU time: ~3 min
    in all; --cpu-short = 1 warm-up + 3 timed at B=16, stated in `sample`)."""
    import numpy as np
    import torch
    from lass_amd import synthetic
    from oracle import resunet as orr
    cores = host_cores()
    torch.set_num_threads(cores)
    osd = orr.to_torch(sd)

    def run(batch, warm, timed):
        _, mix = synthetic.make_mixtures(min(batch, 2), length)
        mix = np.concatenate([mix] * ((batch + 1) // 2))[:batch]
        inp = {"mixture": torch.from_numpy(mix)[:, None, :],
               "condition": torch.from_numpy(synthetic.make_condition(batch))}
        for _ in range(warm):
            orr.forward(osd, inp)
        ts = []
        for _ in range(timed):
            t0 = time.perf_counter()
            orr.forward(osd, inp)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), ts

    t1, ts1 = run(1, 3, 10)
    w16, n16 = (3, 10) if full else (1, 3)
    t16, ts16 = run(16, w16, n16)
    return {"value": 1.0 / t1, "unit": "clips/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "batch16_value": 16.0 / t16,
            "sample": f"oracle/resunet.py (torch-CPU fp32, FFT-based STFT) on {length / 16000:.0f} s clips, "
                      f"{torch.get_num_threads()} threads: B=1 3 warm-up + 10 timed forwards, median {t1:.3f} s "
                      f"(min {min(ts1):.3f}, max {max(ts1):.3f}); B=16 {w16} warm-up + {n16} timed, median {t16:.2f} s "
                      f"(min {min(ts16):.2f}, max {max(ts16):.2f})"
                      + ("" if full else "; SURVEY 8(d) asks 3 + 10 at B=16 too: shortened by --cpu-short")}


def conv_flops(rows, B, wino, wino4=frozenset()):
    """(algorithmic, executed) FLOPs per step of the conv3x3_mfma class (3x3 convs + the 1x1 shortcuts fused into them).
    Executed: a Winograd F(2x2,3x3) launch performs 16 instead of 36 multiplies per 2x2 tile and (cin, cout) - 4/9 of
    the direct count; an F(4x4,3x3) launch (row names in `wino4`) 36 instead of 144 per 4x4 tile - 1/4; the shortcut
    (transform domain or direct) performs the direct 1x1 count."""
    alg = exe = 0.0
    for r in rows:
        if r["kind"] == "3x3":
            alg += 2.0 * B * r["macs"]
            f = 0.25 if (wino and r["name"] in wino4) else ((4.0 / 9.0) if (wino and r["h"] % 2 == 0) else 1.0)
            exe += 2.0 * B * r["macs"] * f
        elif r["name"].endswith(".shortcut"):
            alg += 2.0 * B * r["macs"]
            exe += 2.0 * B * r["macs"]
    return alg, exe


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(parent_launch(args, sys.argv[1:]))

    # RCCL and the HIP runtime print banners on the C-level stdout: from here on fd 1 is stderr, and the one JSON line
    # goes to the real stdout through the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    launched = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    ge.build()
    from lass_amd import arch, synthetic
    from lass_amd import dist as ldist
    from lass_amd.metrics import stats_to_db
    from lass_amd.resunet import ResUNet30

    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev if args.backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # The exchange step runs on a real process group at every N: under the launcher its rendezvous comes from the
    # environment; a plain N=1 run makes a single-rank group so the RCCL path executes there too.
    pg_error = None
    try:
        if launched:
            dist.init_process_group(args.backend, **({"device_id": dev} if args.backend == "nccl" else {}))
        else:
            dist.init_process_group(args.backend, init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1,
                                    **({"device_id": dev} if args.backend == "nccl" else {}))
    except Exception as e:  # a single-GPU run still reports its throughput; N>1 cannot continue
        if world > 1:
            raise
        pg_error = f"{type(e).__name__}: {e}"

    ms_workload = args.workload == "multistft"
    rate = 32000.0 if ms_workload else 16000.0
    if ms_workload:
        from lass_amd.resunet_with_multistft import ResUNet30 as MultiSTFT
        sd = synthetic.make_state_dict_ms()
        model = MultiSTFT(1, 1, 512)
    else:
        sd = synthetic.make_state_dict()
        model = ResUNet30(1, 1, 512)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model = model.to(dev).eval().set_compute_dtype(args.dtype)
    B, L = args.batch, args.length
    # distinct synthetic clips per rank; a small pool tiled to B keeps host-side generation short
    pool = min(B, 4)
    src_np, mix_np = synthetic.make_mixtures(pool, L, first=rank * pool)
    tile = lambda a: np.concatenate([a] * ((B + pool - 1) // pool))[:B]  # noqa: E731
    source = torch.from_numpy(tile(src_np)).to(dev)
    mixture = torch.from_numpy(tile(mix_np)).to(dev)
    cond = torch.from_numpy(synthetic.make_condition(B)).to(dev)
    out = torch.empty_like(mixture)

    def barrier():
        if dist.is_initialized() and world > 1:
            dist.barrier()

    def timed(eng, steps, warmup, io=None):
        """W untimed steps, then exactly `steps` bracketed by barrier + synchronize; max over ranks."""
        mixture_, cond_, out_ = io or (mixture, cond, out)
        for _ in range(warmup):
            eng.separate(mixture_, cond_, out_)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.separate(mixture_, cond_, out_)
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if dist.is_initialized() and world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def profiled(eng, steps, io=None):
        """Second loop with HIP events around every kernel class (on the launch stream) -> {class: (ms, launches)}."""
        mixture_, cond_, out_ = io or (mixture, cond, out)
        eng.set_profiling(True)
        eng.profile(reset=True)
        for _ in range(steps):
            eng.separate(mixture_, cond_, out_)
            torch.cuda.synchronize()
            eng.profile(reset=False)  # folds this step's events so the pre-sized pool is reused
        prof = eng.profile(reset=True)
        eng.set_profiling(False)
        return prof

    check = os.environ.get("LASS_EXP", "0") == "0"  # diagnostic timing experiments produce garbage on purpose
    eng = model.engine
    # lass_separate captures its hipGraph on the third identical call: with --warmup < 3 the remaining untimed calls are
    # made here, so that the timed loop is replay-only at every W (reported as launch.extra_warmup_for_capture)
    extra_warm = max(0, 3 - args.warmup) if eng.graph_stats()[0] else 0
    dt = timed(eng, args.steps, args.warmup + extra_warm)
    assert not check or torch.isfinite(out).all()
    # the same step launched eagerly (no hipGraph replay): what the HIP-event-instrumented loop below runs on
    eng.set_graph_replay(False)
    dt_eager = min(timed(eng, args.steps, 1), timed(eng, args.steps, 1))  # (the faster of two passes: a host hiccup shows at 40 launches per step)
    eng.set_graph_replay(True)

    # ---- the one exchange step: per-clip metric rows, all-gathered over the process group ---------------
    exch = {"rccl_ranks": 0, "allgather_ms": None, "backend": args.backend}
    if pg_error is None:
        st_sep = eng.sdr_stats(source, out).cpu().numpy()
        st_mix = eng.sdr_stats(source, mixture).cpu().numpy()
        sdr, sisdr = stats_to_db(st_sep, L)
        sdr0, _ = stats_to_db(st_mix, L)
        rows_local = np.stack([sdr, sdr - sdr0, sisdr], axis=1)
        ldist.gather_rows(rows_local, world * B, dev)  # first call sets the communicator up
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        allrows = ldist.gather_rows(rows_local, world * B, dev)
        torch.cuda.synchronize()
        exch.update(rccl_ranks=dist.get_world_size() if args.backend == "nccl" else 0, ranks=dist.get_world_size(),
                    allgather_ms=(time.perf_counter() - t0) * 1e3, rows=int(allrows.shape[0]),
                    mean_sdr=float(allrows[:, 0].mean()), mean_sdri=float(allrows[:, 1].mean()),
                    mean_sisdr=float(allrows[:, 2].mean()))
        assert allrows.shape == (world * B, 3) and (not check or np.isfinite(allrows).all())
        assert not check or np.array_equal(allrows[rank * B:(rank + 1) * B], rows_local)
    else:
        exch["error"] = pg_error

    graph_on, graph_caps, graph_replays = eng.graph_stats()
    psteps = max(1, min(args.profile_steps, args.steps))
    prof = profiled(eng, psteps)

    rows = (arch.ms_conv_layer_table if ms_workload else arch.conv_layer_table)(arch.padded_frames(arch.frames_for(L)))
    wino = os.environ.get("LASS_WINO", "1") != "0"
    w4thr = int(os.environ.get("LASS_WINO4", "32"))
    w4rows = arch.wino4_routed(rows, w4thr) if wino else set()

    def mode_record(dtype, dt_mode, steps, prof_mode):
        """Roofline numbers of one compute mode from its timed and profiled loops."""
        alg, exe_f32 = conv_flops(rows, B, wino, w4rows)
        ms, launches = prof_mode["conv3x3_mfma"]
        per_step = ms / psteps * 1e-3
        if dtype == "f32":
            exe, peak = exe_f32, PEAK_F32_MFMA_TFLOPS
        else:  # direct bf16 MFMA convolution; the split mode issues three MFMAs per product
            exe, peak = alg * (3.0 if dtype == "bf16x3" else 1.0), PEAK_BF16_MFMA_TFLOPS
        # (the byte model is ResUNet30's layer table; for the multi-STFT workload the MFMA roofline alone is reported)
        byt = None if ms_workload else float(B) * arch.conv3x3_bytes_per_clip(L, 2 if dtype == "bf16" else 4)
        return {"clips_s": world * B * steps / dt_mode, "ms_per_step": dt_mode / steps * 1e3,
                "conv_ms": per_step * 1e3, "launches_per_step": launches / psteps,
                "avg_launch_ms": ms / max(1, launches),
                "executed_tflops": exe / per_step / 1e12, "algorithmic_tflops": alg / per_step / 1e12,
                "peak_tflops": peak, "frac": exe / per_step / 1e12 / peak,
                "hbm_algorithmic_gbs": None if byt is None else byt / per_step / 1e9,
                "hbm_frac": None if byt is None else byt / per_step / 1e9 / PEAK_HBM_GBS,
                "hbm_model": ("blocked bf16 activation storage between conv launches (2 B/element)" if dtype == "bf16"
                              else "f32 activation storage (4 B/element)") + ", ideal per-launch fusion (SURVEY 8d)",
                "kernel_ms_per_step": {k: v[0] / psteps for k, v in prof_mode.items()}}

    head = mode_record(args.dtype, dt, args.steps, prof)

    def alg_bytes_launch_of(rec):
        if ms_workload:
            return None
        return float(B) * arch.conv3x3_bytes_per_clip(L, 2 if args.dtype == "bf16" else 4) / max(1.0, rec["launches_per_step"])

    modes = {}
    want = args.modes
    if want == "auto":
        want = "bf16,bf16x3,multistft,evaluator" if (world == 1 and args.dtype == "f32" and not ms_workload) else "none"
    if ms_workload and any(x in want for x in ("multistft", "evaluator")):
        raise SystemExit("--workload multistft: --modes takes bf16 / bf16x3 / none only")
    want = [x for x in want.split(",") if x and x != "none"]
    for m in [x for x in want if x not in ("multistft", "evaluator")]:
        model.set_compute_dtype(m)
        e2 = model.engine
        # >= 3 untimed steps: lass_separate captures its hipGraph on the third identical call, which must not fall inside
        # the timed region
        c0, r0 = e2.graph_stats()[1:]
        dt_m = timed(e2, args.steps, max(3, args.warmup))
        assert not check or torch.isfinite(out).all()
        c1, r1 = e2.graph_stats()[1:]
        modes[m] = mode_record(m, dt_m, args.steps, profiled(e2, psteps))
        modes[m]["launch"] = {"captures": c1 - c0, "replays": r1 - r0, "half_batch_overlap": half_batch_overlap(m, B)}
        if not ms_workload:
            alg_b = float(B) * arch.conv3x3_bytes_per_clip(L, 2 if m == "bf16" else 4) / max(1.0, modes[m]["launches_per_step"])
            modes[m]["algorithmic_bytes_per_launch"] = alg_b
            modes[m].update(traffic_fields(m, alg_b, (B, L) == (16, 160000)))
    if modes:
        model.set_compute_dtype(args.dtype)

    if "multistft" in want:
        # BASELINE configs[4] on one GPU: the multi-resolution-STFT separator (models/resunet_with_multistft.py:137-216
        # under the authored spec of DESIGN.md section 9) on ONE 30 s @ 32 kHz clip (L = 960 000, 6016 x 1024 bins),
        # 3 warm-up + 5 timed steps, then 2 profiled steps for the conv3x3 class time.
        from lass_amd.resunet_with_multistft import ResUNet30 as MultiSTFT
        import gc
        Lm, Bm = 960000, 1  # its 7.4-GiB workspace sits beside the headline context's 7.8 GB: 288 GB of HBM
        ms = MultiSTFT(1, 1, 512)
        ms.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict_ms().items()})
        ms = ms.to(dev).eval()
        em = ms.engine
        mix_m = torch.from_numpy(synthetic.make_mixtures(Bm, Lm, first=rank)[1]).to(dev)
        io = (mix_m, torch.from_numpy(synthetic.make_condition(Bm)).to(dev), torch.empty_like(mix_m))
        dt_m = timed(em, 5, 3, io)
        assert not check or torch.isfinite(io[2]).all()
        pm = profiled(em, 2, io)
        rows_m = arch.ms_conv_layer_table(arch.padded_frames(arch.frames_for(Lm)))
        alg_m, exe_m = conv_flops(rows_m, Bm, wino, arch.wino4_routed(rows_m, w4thr) if wino else set())
        cls_s = pm["conv3x3_mfma"][0] / 2 * 1e-3
        tot_m = 2.0 * Bm * sum(r["macs"] for r in rows_m)
        modes["multistft_30s_32k"] = {
            "workload": "multi-STFT ResUNet30 (windows 256/512/2048 at n_fft 2048, authored spec) f32, B=1, ONE 30 s @ "
                        "32 kHz clip (BASELINE configs[4] per GPU); 3 warm-up + 5 timed steps",
            "ms_per_clip": dt_m / 5 / Bm * 1e3, "clips_s": world * Bm * 5 / dt_m,
            "realtime_factor": world * Bm * 5 / dt_m * 30.0, "dtype": "f32",
            "conv_ms": cls_s * 1e3, "launches_per_step": pm["conv3x3_mfma"][1] / 2,
            "algorithmic_tflops": alg_m / cls_s / 1e12, "executed_tflops": exe_m / cls_s / 1e12,
            "peak_tflops": PEAK_F32_MFMA_TFLOPS, "frac": exe_m / cls_s / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "whole_step_tflops": tot_m / (dt_m / 5) / 1e12, "gmac_per_clip": tot_m / 2e9 / Bm,
            "workspace_gib": em.workspace_bytes(Bm, Lm) / 2.0 ** 30,
            "kernel_ms_per_step": {k: v[0] / 2 for k, v in pm.items()}}
        # the same clip in the bf16 compute modes (f32 tensors between the blocks): 3 warm-up + 5 timed steps each
        for mm in ("bf16x3", "bf16"):
            ms.set_compute_dtype(mm)
            em = ms.engine
            dt_b = timed(em, 5, 3, io)
            assert not check or torch.isfinite(io[2]).all()
            modes["multistft_30s_32k"][mm] = {"ms_per_clip": dt_b / 5 / Bm * 1e3, "clips_s": world * Bm * 5 / dt_b,
                                               "realtime_factor": world * Bm * 5 / dt_b * 30.0}
        del em, ms, io, mix_m
        gc.collect()
        torch.cuda.empty_cache()

    if "evaluator" in want:
        # SURVEY 8(f1), driver-timed: DCASEEvaluator.__call__ (dcase_evaluator.py:49-122) end to end on a synthetic validation set
        # of 260 clips (16 full batches + a ragged tail of 4): WAV decode on prefetch threads -> pinned staging -> H2D -> device-side
        # mixing at SNR -> lass_separate -> device-side SDR / SI-SDR -> means; 1 warm-up call, median of 5 timed calls, against the
        # headline separator rate of this same run
        import shutil
        import tempfile
        from lass_amd.audiosep import AudioSep, PrecomputedQueryEncoder
        from lass_amd.evaluator import DCASEEvaluator
        n_ev = 260
        tmp = tempfile.mkdtemp(prefix="lass_eval_")
        try:
            csv_path = synthetic.write_validation_set(tmp, n_clips=n_ev, length=L)
            plm = AudioSep(ss_model=model, query_encoder=PrecomputedQueryEncoder())
            ev = DCASEEvaluator(16000, csv_path, os.path.join(tmp, "lass_validation"), batch_size=B)
            ev(plm)
            dts, outv = [], None
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                outv = ev(plm)
                torch.cuda.synchronize()
                dts.append(time.perf_counter() - t0)
            dt_ev = sorted(dts)[2]
            sep_rate = world * B * args.steps / dt
            modes["evaluator_e2e"] = {
                "workload": f"DCASEEvaluator.__call__ on {n_ev} synthetic 10 s mixtures (WAV files -> decode -> mix at SNR -> separate "
                            f"-> SDR/SDRi/SI-SDR means), batch {B}, ragged tail of {n_ev % B}; 1 warm-up call, median of 5 (the decode threads share the box's CPUs: single calls scatter by 20 %)",
                "clips_s": n_ev / dt_ev, "calls_s": [n_ev / d for d in dts], "separator_clips_s": sep_rate,
                "evaluator_over_separator": n_ev / dt_ev / sep_rate,
                "data_path": ev.last_path, "resident_batches": ev.resident_batches, "generic_batches": ev.generic_batches,
                "mean_sisdr_sdri_sdr": [float(v) for v in outv]}
            if "bf16" in modes:
                # the same call with the bf16-MFMA separator (configs[2]): at ~3 300 clips/s the loop's own work - 20 MB of H2D,
                # the mixing and statistics kernels, 32 file reads per batch - is a visible share; reported, not hidden
                model.set_compute_dtype("bf16")
                ev(plm)
                dtb = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    outb = ev(plm)
                    torch.cuda.synchronize()
                    dtb.append(time.perf_counter() - t0)
                dt_b = sorted(dtb)[2]
                modes["evaluator_e2e_bf16"] = {
                    "clips_s": n_ev / dt_b, "calls_s": [n_ev / d for d in dtb], "separator_clips_s": modes["bf16"]["clips_s"],
                    "evaluator_over_separator": n_ev / dt_b / modes["bf16"]["clips_s"], "data_path": ev.last_path,
                    "mean_sisdr_sdri_sdr": [float(v) for v in outb]}
                model.set_compute_dtype(args.dtype)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    if rank == 0:
        alg_step, exe_step = conv_flops(rows, B, wino, w4rows)
        total_flops = 2.0 * B * sum(r["macs"] for r in rows)
        traffic = traffic_fields(args.dtype, alg_bytes_launch_of(head), (B, L) == (16, 160000) and not ms_workload)
        alg_bytes_launch = alg_bytes_launch_of(head)
        kernel = {"f32": (f"wino4_kernel<...> x{len(w4rows)} (Winograd F(4x4,3x3): {', '.join(sorted(w4rows))}) + wino32_kernel / "
                          f"wino_kernel x{26 - len(w4rows)} (F(2x2,3x3): the 16-/8-bin layers) on "
                          "v_mfma_f32_16x16x4_f32, 1x1 shortcuts fused (conv3x3_mfma class)") if wino else "conv_kernel (direct f32 MFMA)",
                  "bf16": "conv_bf16_kernel x22 + enc1_fused / dec6u_fused_bf16_kernel (a ConvBlockRes each, decoder_block6's with its "
                          "transposed conv inside): direct 3x3 + fused 1x1 shortcuts on v_mfma_f32_32x32x16_bf16",
                  "bf16x3": "conv_bf16_kernel (split operands, 3 MFMAs per product) x26"}[args.dtype]
        res = {
            "metric": "clips/sec (30s@32kHz)" if ms_workload else "clips/sec (10s@16kHz)", "value": world * B * args.steps / dt, "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"multi-STFT ResUNet30 (windows 256/512/2048 at n_fft 2048, authored spec: DESIGN.md section 9) "
                                    f"{'fp32' if args.dtype == 'f32' else args.dtype + ' MFMA convolutions'}, batch={B}/GPU, "
                                    f"{L / rate:.0f}s@32kHz clips, seeded random-init weights (BASELINE configs[4] per GPU; a clip "
                                    f"here is {L / rate:.0f} s, not the metric's 10 s: see realtime_factor)") if ms_workload else
                                   (f"ResUNet30 separate() {'fp32' if args.dtype == 'f32' else args.dtype + ' MFMA convolutions'}, "
                                    f"batch={B}/GPU, {L / 16000:.0f}s@16kHz clips, fixed precomputed condition embedding, "
                                    f"seeded random-init weights (BASELINE configs[{1 if args.dtype == 'f32' else 2}]"
                                    f"{'; configs[3] sharding' if world > 1 else ''})"),
                       "clips_per_gpu_per_step": B, "samples_per_clip": L, "sample_rate": int(rate),
                       "parallelism": f"clip-sharded x{world}"},
            "realtime_factor": world * B * args.steps / dt * (L / rate),
            "exchange": exch,
            "launch": {"hipgraph_replay": graph_on, "captures": graph_caps, "replays_in_headline_loops": graph_replays,
                       "half_batch_overlap": half_batch_overlap(args.dtype, B),
                       "extra_warmup_for_capture": extra_warm,
                       "eager_ms_per_step": dt_eager / args.steps * 1e3,
                       "profiled_ms_per_step": sum(v for v in head["kernel_ms_per_step"].values()),
                       "note": "ms_per_step / value: the timed loop replays ONE captured hipGraph of the ~40 launches per "
                               "step; eager_ms_per_step: the same loop with replay switched off; kernel_ms_per_step / "
                               "class_ms_per_step: a third, eager loop with HIP events around every kernel class (their sum = "
                               "profiled_ms_per_step: events add the drain between classes).  half_batch_overlap: the replayed "
                               "graph runs the step as two half-batches on two branches (LASS_SPLIT; DESIGN.md 5b); the eager and "
                               "the profiled loop launch it unsplit"},
            "roofline": {"bound": "mfma", "kernel": kernel,
                         "achieved": head["executed_tflops"], "peak": head["peak_tflops"], "unit": "TFLOP/s",
                         "frac": head["frac"],
                         "achieved_is": "EXECUTED MFMA FLOPs of the class per step / its HIP-event time per step "
                                        "(profiled loop); the algorithmic direct-conv rate is algorithmic_tflops",
                         "algorithmic_tflops": head["algorithmic_tflops"],
                         "winograd_mult_reduction": (alg_step / exe_step) if (args.dtype == "f32" and wino) else 1.0,
                         "winograd_f4x4_layers": sorted(w4rows) if args.dtype == "f32" else [],
                         "algorithmic_gflop_per_step": alg_step / 1e9,
                         **traffic,
                         "algorithmic_bytes_per_launch": alg_bytes_launch,
                         "launches_per_step": head["launches_per_step"], "avg_launch_ms": head["avg_launch_ms"],
                         "class_ms_per_step": head["conv_ms"], "profiled_steps": psteps,
                         "hbm_algorithmic_gbs": head["hbm_algorithmic_gbs"], "hbm_peak_gbs": PEAK_HBM_GBS,
                         "hbm_frac": head["hbm_frac"], "hbm_model": head["hbm_model"],
                         "whole_step_tflops": total_flops / (dt / args.steps) / 1e12},
            "kernel_ms_per_step": head["kernel_ms_per_step"],
        }
        if modes:
            res["modes"] = modes
        if world == 1 and not args.no_cpu_baseline and args.dtype == "f32":
            res["cpu_baseline"] = cpu_baseline(sd, L, not args.cpu_short)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
