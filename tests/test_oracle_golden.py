"""Pin the oracle (oracle/resunet.py) to the reference's own models/resunet.py via tests/golden/*.npz.

The fixtures were produced by tools/gen_golden.py, which imports and runs the reference file itself.  Inputs and
weights are regenerated from the seed here.  CPU only.
"""
import json
import os

import numpy as np
import pytest
import torch

from lass_amd import arch, synthetic
from oracle import metrics as om
from oracle import resunet as orr


def _sample(t, n=4096):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().astype(np.float32)


def _stats(t):
    d = t.detach().double()
    return np.asarray([d.sum().item(), d.abs().sum().item(), (d * d).sum().item(), float(d.numel())])


def test_param_specs_match_reference_state_dict(golden_dir):
    spec = json.load(open(os.path.join(golden_dir, "state_dict_spec.json")))
    ours = {n: list(s) for n, s, _ in arch.param_specs()}
    theirs = {k: v["shape"] for k, v in spec.items() if not k.startswith(("base.stft.", "base.istft."))}
    assert ours == theirs
    # the only extra keys in a reference checkpoint are torchlibrosa's frozen DFT buffers
    extra = sorted(k for k in spec if k not in ours)
    assert all(k.startswith(("base.stft.", "base.istft.")) for k in extra)
    n_learn = sum(int(np.prod(s)) for n, s, k in arch.param_specs() if k not in ("bn_mean", "bn_var", "bn_nbt"))
    assert n_learn == 26446917


@pytest.fixture(scope="module")
def g1_run(synthetic_sd):
    sd = orr.to_torch(synthetic_sd)
    _, mix = synthetic.make_mixtures(2, 16000)
    cond = synthetic.make_condition(2)
    taps = {}
    out = orr.forward(sd, {"mixture": torch.from_numpy(mix)[:, None], "condition": torch.from_numpy(cond)},
                      taps, stft_form="dft")
    return sd, cond, taps, out["waveform"]


def test_g1_waveform(golden_dir, g1_run):
    g = np.load(os.path.join(golden_dir, "g1_tiny.npz"))
    _, _, _, wav = g1_run
    ref = g["waveform"]
    err = np.sqrt(np.mean((wav.numpy() - ref) ** 2))
    assert err <= 1e-6 * max(1.0, np.sqrt(np.mean(ref ** 2)) / 0.04), err
    assert np.max(np.abs(wav.numpy() - ref)) < 2e-5


def test_g1_film(golden_dir, g1_run):
    g = np.load(os.path.join(golden_dir, "g1_tiny.npz"))
    sd, cond, _, _ = g1_run
    ours = orr.film_all(sd, torch.from_numpy(cond))
    keys = [k[len("film/"):] for k in g.files if k.startswith("film/")]
    assert sorted(keys) == sorted(ours) and len(keys) == 38
    for k in keys:
        np.testing.assert_allclose(ours[k].numpy(), g["film/" + k], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", ["pre", "encoder_block1", "encoder_block1.pool", "encoder_block3", "encoder_block6.pool",
                                  "conv_block7a", "decoder_block1.up", "decoder_block1", "decoder_block4",
                                  "decoder_block6.up", "decoder_block6", "out_real", "out_imag"])
def test_g1_taps(golden_dir, g1_run, name):
    g = np.load(os.path.join(golden_dir, "g1_tiny.npz"))
    _, _, taps, _ = g1_run
    t = taps[name]
    scale = max(1.0, float(np.sqrt(g[name + "/stats"][2] / g[name + "/stats"][3])))
    np.testing.assert_allclose(_sample(t), g[name + "/sample"], rtol=0, atol=3e-4 * scale)
    st = _stats(t)
    assert st[3] == g[name + "/stats"][3]
    np.testing.assert_allclose(st[:3], g[name + "/stats"][:3], rtol=2e-5, atol=1e-2)


def test_g1_x_center_full(golden_dir, g1_run):
    g = np.load(os.path.join(golden_dir, "g1_tiny.npz"))
    _, _, taps, _ = g1_run
    np.testing.assert_allclose(taps["conv_block7a.pool"].numpy(), g["x_center"], rtol=0, atol=2e-4)


def test_g2_clip10s(golden_dir, synthetic_sd):
    g = np.load(os.path.join(golden_dir, "g2_clip10s.npz"))
    sd = orr.to_torch(synthetic_sd)
    src, mix = synthetic.make_mixtures(1, 160000, first=3)
    cond = synthetic.make_condition(1)
    w = orr.forward(sd, {"mixture": torch.from_numpy(mix)[:, None], "condition": torch.from_numpy(cond)})[
        "waveform"][0, 0].numpy()
    assert np.sqrt(np.mean((w[::16] - g["waveform_dec"]) ** 2)) < 2e-6
    np.testing.assert_allclose(w[:2048], g["head"], atol=2e-5)
    np.testing.assert_allclose(w[-2048:], g["tail"], atol=2e-5)
    np.testing.assert_allclose(np.sqrt((w.reshape(160, 1000) ** 2).mean(1)), g["rms_1k"], rtol=1e-4, atol=1e-6)
    sdr = om.calculate_sdr(src[0], w)
    sdr0 = om.calculate_sdr(src[0], mix[0])
    sisdr = om.calculate_sisdr(src[0], w)
    np.testing.assert_allclose([sdr, sdr - sdr0, sisdr], g["sdr_triple"], atol=0.01)  # << +-0.05 dB of north_star


def test_g3_chunk_inference(golden_dir, synthetic_sd):
    g = np.load(os.path.join(golden_dir, "g3_chunk.npz"))
    sd = orr.to_torch(synthetic_sd)
    segs = [synthetic.make_mixtures(1, 160000, first=10 + i)[1][0] for i in range(3)]
    long_mix = np.concatenate(segs)[:400000].astype(np.float32)
    cond = synthetic.make_condition(1)
    out = orr.chunk_inference(sd, {"mixture": torch.from_numpy(long_mix)[None, None, :],
                                   "condition": torch.from_numpy(cond)})
    assert out.shape == (1, 400000) and out.dtype == np.float64
    assert np.sqrt(np.mean((out[0, ::25] - g["out_dec"]) ** 2)) < 2e-6
    for s, seam in zip((32000, 128000, 224000, 320000), g["seams"]):
        np.testing.assert_allclose(out[0, s - 64:s + 64], seam, atol=2e-5)
    # short input: the loop never runs and zeros come back (resunet.py:682)
    short = orr.chunk_inference(sd, {"mixture": torch.zeros(1, 1, 160000), "condition": torch.from_numpy(cond)})
    assert short.shape == (1, 160000) and not short.any()


# ---- G4 (round 4): a second seeded weight set / condition / clips / length, every element of the taps counted ------------
G4_SEED = synthetic.SEED + 17
G4_TAPS = ("encoder_block1", "encoder_block3", "decoder_block3", "decoder_block5", "decoder_block6.up", "out_real", "out_imag")


def _margins(t):
    d = t.detach().double()
    return d.sum(3).numpy(), d.sum(2).numpy()


def check_against_g4(g, waveform, taps, tol_scale=1.0):
    """Shared by the CPU (oracle) and GPU (HIP path) tests: waveform in full, decoder_block1 in full, row / column margins of
    the larger taps (every element lands in one row sum and one column sum: an isolated wrong pixel cannot hide)."""
    ref = g["waveform"]
    err = float(np.sqrt(np.mean((waveform - ref) ** 2)))
    assert err <= 3e-6 * tol_scale, err                      # north_star bar: 1e-4 RMS
    assert float(np.max(np.abs(waveform - ref))) < 1e-4 * tol_scale
    d1 = taps["decoder_block1"]
    assert d1.shape == g["decoder_block1"].shape
    scale = float(np.sqrt(np.mean(g["decoder_block1"] ** 2)))
    assert float(np.max(np.abs(d1 - g["decoder_block1"]))) < 3e-4 * scale * tol_scale
    for n in G4_TAPS:
        t = taps[n]
        rows, cols = _margins(torch.from_numpy(np.ascontiguousarray(t)))
        rms = float(np.sqrt(g[n + "/stats"][2] / g[n + "/stats"][3]))
        for got, want, cnt in ((rows, g[n + "/rows"], t.shape[3]), (cols, g[n + "/cols"], t.shape[2])):
            assert got.shape == want.shape, n
            # a sum of cnt elements of typical size rms: rounding noise ~ 1e-6 * rms * sqrt(cnt); bound 2e-4 * rms * sqrt(cnt)
            assert float(np.max(np.abs(got - want))) < 2e-4 * rms * np.sqrt(cnt) * tol_scale, n


def test_g4_second_weight_set(golden_dir):
    """The oracle against the reference's own output for ANOTHER seeded weight set, conditions, clips and length (G4;
    tools/gen_golden.py --only g4): the pin of the oracle does not rest on one set of weights."""
    g = np.load(os.path.join(golden_dir, "g4_second_weights.npz"))
    sd = orr.to_torch(synthetic.make_state_dict(seed=G4_SEED))
    _, mix = synthetic.make_mixtures(2, 48000, first=20)
    cond = synthetic.make_condition(2, seed=G4_SEED)
    taps = {}
    out = orr.forward(sd, {"mixture": torch.from_numpy(mix)[:, None], "condition": torch.from_numpy(cond)}, taps, stft_form="dft")
    got = {}
    for n in G4_TAPS + ("decoder_block1",):
        v = taps[n]
        got[n] = (v[1] if isinstance(v, tuple) else v).numpy()
    check_against_g4(g, out["waveform"].numpy(), got)
