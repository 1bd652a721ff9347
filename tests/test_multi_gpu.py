"""The N > 1 path (BASELINE configs[3]): clip-sharded evaluation with ONE all-gather of per-clip metric rows,
and bench.py's self-launch of one rank per GPU.

CPU (`-m "not gpu"`): the launcher logic.  GPU (`-m gpu`): two rank processes sharing cuda:0 over gloo run
DCASEEvaluator.__call__ and bench.py end to end; a single-rank run exercises the RCCL ("nccl") exchange itself."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "2"
    return env


def test_bench_parent_builds_launch_command_without_touching_torch():
    """`python bench.py --gpus N` with no WORLD_SIZE: the parent only assembles the torch.distributed.run command
    (one fresh rank per GPU, rendezvous on 127.0.0.1) - it must not import torch, let alone initialise a GPU."""
    env = _env()
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-X", "importtime", BENCH, "--gpus", "8", "--steps", "3", "--warmup", "1",
                        "--print-launch"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"] or "--print-launch" in cmd
    assert os.path.samefile(cmd[cmd.index("--master-port") + 2], BENCH)
    imported = [ln.split("|")[-1].strip() for ln in r.stderr.splitlines() if ln.startswith("import time:")]
    assert "torch" not in imported and "numpy" not in imported


def test_bench_workload_switch_defaults():
    """`--workload multistft` (configs[4]'s per-rank job) changes the defaults of --batch / --length only; the launch command the
    parent assembles carries the switch to every rank."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    a = bench.parse_args([])
    assert (a.workload, a.batch, a.length, a.gpus, a.dtype) == ("resunet30", 16, 160000, 1, "f32")
    m = bench.parse_args(["--workload", "multistft"])
    assert (m.batch, m.length) == (1, 960000)
    m2 = bench.parse_args(["--workload", "multistft", "--batch", "2", "--length", "320000"])
    assert (m2.batch, m2.length) == (2, 320000)
    cmd = bench.launch_command(["--gpus", "8", "--workload", "multistft"], 8, 12345)
    assert cmd[-4:] == ["--gpus", "8", "--workload", "multistft"] and "--nproc-per-node=8" in cmd


def test_bench_rank_refuses_mismatched_world():
    env = _env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


_EVAL_WORKER = r"""
import os, sys
root, csv_path, audio_dir, out_dir = sys.argv[1:5]
sys.path.insert(0, root)
import numpy as np, torch, torch.distributed as dist
from lass_amd import synthetic
from lass_amd.audiosep import AudioSep, PrecomputedQueryEncoder
from lass_amd.evaluator import DCASEEvaluator
from lass_amd.resunet import ResUNet30
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
sd = synthetic.make_state_dict()
m = ResUNet30(1, 1, 512)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
m = m.to("cuda:0").eval()
pl_model = AudioSep(ss_model=m, query_encoder=PrecomputedQueryEncoder())
ev = DCASEEvaluator(sampling_rate=16000, eval_indexes=csv_path, audio_dir=audio_dir,
                    batch_size=int(sys.argv[5]) if len(sys.argv) > 5 else 3)
res = ev(pl_model)
np.save(os.path.join(out_dir, f"rows_rank{rank}.npy"), ev.last_rows)
np.save(os.path.join(out_dir, f"means_rank{rank}.npy"), np.asarray(res))
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_sharded_evaluator_two_ranks_equals_single_rank(tmp_path):
    """dcase_evaluator.py:65-122 sharded over 2 rank processes (both on cuda:0, backend gloo): the gathered per-clip
    rows and the three means equal the world-size-1 run.  7 clips -> ragged shards (3 + 4) and NaN padding."""
    from lass_amd import synthetic
    from lass_amd.audiosep import AudioSep, PrecomputedQueryEncoder
    from lass_amd.evaluator import DCASEEvaluator
    from lass_amd.resunet import ResUNet30
    n, L = 7, 24000
    csv_path = synthetic.write_validation_set(str(tmp_path), n_clips=n, length=L)
    audio_dir = os.path.join(str(tmp_path), "lass_validation")
    script = os.path.join(str(tmp_path), "w.py")
    open(script, "w").write(_EVAL_WORKER)
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), script, ROOT, csv_path, audio_dir, str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    sd = synthetic.make_state_dict()
    m = ResUNet30(1, 1, 512)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.to("cuda:0").eval()
    ev = DCASEEvaluator(sampling_rate=16000, eval_indexes=csv_path, audio_dir=audio_dir, batch_size=3)
    single = np.asarray(ev(AudioSep(ss_model=m, query_encoder=PrecomputedQueryEncoder())))
    for rank in (0, 1):
        rows = np.load(os.path.join(str(tmp_path), f"rows_rank{rank}.npy"))
        means = np.load(os.path.join(str(tmp_path), f"means_rank{rank}.npy"))
        assert rows.shape == (n, 3) and not np.isnan(rows).any()
        # same kernels on the same clips; only the batch composition differs (3+... vs 3+3+1), which the kernels do
        # not see (eval-mode BN: no cross-clip coupling) -> bit-identical rows
        np.testing.assert_allclose(rows, ev.last_rows, rtol=0, atol=1e-9)
        np.testing.assert_allclose(means, single, rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_configs3_workload_128_clips_two_ranks(tmp_path):
    """BASELINE configs[3] at its stated size: 128 clips of 10 s @ 16 kHz through DCASEEvaluator.__call__
    (dcase_evaluator.py:65-122) in 16-clip batches, clip-sharded over 2 rank processes (64 clips = four batches each; both
    on cuda:0, backend gloo - RCCL refuses two ranks on one device), one all-gather of the metric rows.  The gathered rows
    and means equal the world-size-1 run, and 4 sampled clips equal the CPU oracle's evaluator to 0.01 dB."""
    from lass_amd import synthetic
    from lass_amd.audiosep import AudioSep, PrecomputedQueryEncoder
    from lass_amd.evaluator import DCASEEvaluator
    from lass_amd.resunet import ResUNet30
    from oracle import evaluator as oev
    from oracle import resunet as orr
    n, L = 128, 160000
    csv_path = synthetic.write_validation_set(str(tmp_path), n_clips=n, length=L)
    audio_dir = os.path.join(str(tmp_path), "lass_validation")
    script = os.path.join(str(tmp_path), "w.py")
    open(script, "w").write(_EVAL_WORKER)
    port = 31500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), script, ROOT, csv_path, audio_dir, str(tmp_path), "16"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env())
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    sd = synthetic.make_state_dict()
    m = ResUNet30(1, 1, 512)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.to("cuda:0").eval()
    qe = PrecomputedQueryEncoder()
    ev = DCASEEvaluator(sampling_rate=16000, eval_indexes=csv_path, audio_dir=audio_dir, batch_size=16)
    single = np.asarray(ev(AudioSep(ss_model=m, query_encoder=qe)))
    assert ev.last_rows.shape == (n, 3) and np.isfinite(ev.last_rows).all()
    for rank in (0, 1):
        rows = np.load(os.path.join(str(tmp_path), f"rows_rank{rank}.npy"))
        means = np.load(os.path.join(str(tmp_path), f"means_rank{rank}.npy"))
        assert rows.shape == (n, 3) and not np.isnan(rows).any()
        np.testing.assert_allclose(rows, ev.last_rows, rtol=0, atol=1e-9)
        np.testing.assert_allclose(means, single, rtol=0, atol=1e-9)
    # the CPU oracle's evaluator on clips from both shards (first / last batch of each)
    picks = [0, 37, 64, 127]
    clips = [synthetic.make_clip(i, L) for i in picks]
    conds = qe.get_query_embed(modality="text", text=[f"synthetic tone cluster {i % 4}" for i in picks]).numpy()
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    _, orows = oev.evaluate(orr.to_torch(sd), clips, conds)
    np.testing.assert_allclose(ev.last_rows[picks], orows, rtol=0, atol=0.01)


def _bench_json(args, timeout=900):
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_self_launches_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` exactly as the driver calls it (no launcher around it): two fresh ranks, one JSON line,
    the exchange step in the job.  gloo because both ranks share the one GPU of the test box (RCCL refuses that)."""
    res = _bench_json(["--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "2",
                       "--length", "32000", "--no-cpu-baseline", "--modes", "none"])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["scaling"] == "weak"
    assert res["exchange"]["ranks"] == 2 and res["exchange"]["rows"] == 4 and res["exchange"]["allgather_ms"] > 0
    assert res["value"] == pytest.approx(2 * 2 * 2 / (res["ms_per_step"] * 2e-3), rel=1e-6)
    assert 0 < res["roofline"]["frac"] <= 1.0


@pytest.mark.gpu
def test_bench_two_ranks_at_the_full_configs3_per_rank_shape():
    """The command the driver scales (`bench.py --gpus N`) at configs[3]'s FULL per-rank job - 16 clips of 10 s @ 16 kHz per
    rank and step - with two ranks (gloo: both share the test box's one GPU; the driver's run uses RCCL, one GPU each).  The
    exchange step gathers 2 x 16 rows, `value` is the whole job's clips over the max-over-ranks time."""
    res = _bench_json(["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "3", "--modes", "none",
                       "--no-cpu-baseline"], timeout=1500)
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["scaling"] == "weak" and res["dtype"] == "f32"
    cfg = res["config"]
    assert cfg["clips_per_gpu_per_step"] == 16 and cfg["samples_per_clip"] == 160000 and "configs[3] sharding" in cfg["workload"]
    ex = res["exchange"]
    assert ex["ranks"] == 2 and ex["rows"] == 32 and ex["allgather_ms"] > 0 and np.isfinite(ex["mean_sdr"])
    assert res["value"] == pytest.approx(2 * 16 * 3 / (res["ms_per_step"] * 3e-3), rel=1e-6)
    # two ranks time-share one GPU here: the whole-job rate stays near the one-GPU rate (each rank sees about half of it);
    # on the driver's node every rank has its own GPU and the same line reads N x the N=1 value minus the exchange
    assert 300 < res["value"] < 1400, res["value"]
    assert 0 < res["roofline"]["frac"] <= 1.0 and "cpu_baseline" not in res and "modes" not in res


@pytest.mark.gpu
def test_bench_multistft_workload_two_ranks():
    """configs[4]'s per-GPU job (multi-STFT ResUNet, ONE 30 s @ 32 kHz clip per rank and step) through the same bench command
    and exchange step: `--workload multistft`, two gloo ranks on the one GPU."""
    res = _bench_json(["--gpus", "2", "--backend", "gloo", "--workload", "multistft", "--steps", "2", "--warmup", "3",
                       "--modes", "none", "--no-cpu-baseline"], timeout=1500)
    assert res["n_gpus"] == 2 and res["metric"] == "clips/sec (30s@32kHz)"
    cfg = res["config"]
    assert cfg["clips_per_gpu_per_step"] == 1 and cfg["samples_per_clip"] == 960000 and cfg["sample_rate"] == 32000
    assert "configs[4]" in cfg["workload"]
    ex = res["exchange"]
    assert ex["ranks"] == 2 and ex["rows"] == 2 and np.isfinite(ex["mean_sisdr"])
    assert res["value"] == pytest.approx(2 * 1 * 2 / (res["ms_per_step"] * 2e-3), rel=1e-6)
    assert res["realtime_factor"] == pytest.approx(res["value"] * 30.0, rel=1e-9)
    assert 0 < res["roofline"]["frac"] <= 1.0 and res["roofline"]["traffic"] is None


@pytest.mark.gpu
def test_bench_single_rank_runs_the_rccl_exchange():
    """N=1: the per-clip metric rows still go through a torch.distributed all_gather on backend "nccl" (= RCCL), so the
    path configs[3] relies on executes on every bench run; roofline.frac is an executed-FLOP fraction (<= 1)."""
    res = _bench_json(["--steps", "2", "--warmup", "1", "--batch", "2", "--length", "32000", "--no-cpu-baseline",
                       "--modes", "bf16"])
    ex = res["exchange"]
    assert ex.get("error") is None and ex["rccl_ranks"] == 1 and ex["rows"] == 2 and np.isfinite(ex["mean_sdr"])
    rf = res["roofline"]
    assert 0 < rf["frac"] <= 1.0 and rf["algorithmic_tflops"] >= rf["achieved"]
    # 9 / 4 with F(2x2,3x3) everywhere; the layers lass_amd.arch.wino4_routed names run as F(4x4,3x3) (4 x fewer multiplies than direct)
    assert 2.25 < rf["winograd_mult_reduction"] < 4.0 and "decoder_block6.conv1" in rf["winograd_f4x4_layers"]
    assert "eager_ms_per_step" in res["launch"] and res["launch"]["eager_ms_per_step"] > 0
    assert set(res["modes"]) == {"bf16"} and 0 < res["modes"]["bf16"]["frac"] <= 1.0
