"""Multi-resolution-STFT separator (BASELINE configs[4]; authored spec: lass_amd/arch.py).

PARITY UNPINNED against the reference: models/resunet_with_multistft.py cannot run and was not imported.
What these tests hold:
  CPU  - the oracle's zero-padded-window STFT / iSTFT equal torch.stft / torch.istft(n_fft=2048, win_length=w); the
         oracle model is built only from oracle/resunet.py blocks that ARE pinned to the reference (tests/golden);
         names / shapes of the module tree follow resunet_with_multistft.py:40-118.
  GPU  - the HIP path (lass_create_multistft) against that oracle: stage level (analysis, synthesis, taps) and end to
         end, tiny shape and the full configs[4] clip (30 s @ 32 kHz, L = 960 000)."""
import numpy as np
import pytest
import torch

from lass_amd import arch, synthetic
from oracle import resunet as orr
from oracle import resunet_multistft as oms

DEV = "cuda:0"
WINS = arch.MS_WIN_LENGTHS


# ---- CPU -----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w", WINS)
def test_oracle_padded_window_stft_equals_torch(w):
    L = 8077
    x = torch.randn(2, L, dtype=torch.float64, generator=torch.Generator().manual_seed(w))
    re, im = oms.stft(x, w)
    ts = torch.stft(x, 2048, 160, w, torch.hann_window(w, periodic=True, dtype=torch.float64), center=True,
                    pad_mode="reflect", normalized=False, onesided=True, return_complex=True).transpose(1, 2)
    assert re.shape == (2, 1, 1 + L // 160, 1025)
    assert float((ts.real - re[:, 0]).abs().max()) < 1e-10 and float((ts.imag - im[:, 0]).abs().max()) < 1e-10


def test_oracle_istft_equals_torch_istft_and_roundtrip():
    L = 9000
    x = torch.randn(2, L, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    re, im = oms.stft(x, 512)
    y = oms.istft(re, im, L, 512)
    assert float((y - x).abs().max()) < 1e-10
    spec = torch.complex(re[:, 0], im[:, 0]).transpose(1, 2)
    ref = torch.istft(spec, 2048, 160, 512, torch.hann_window(512, periodic=True, dtype=torch.float64), center=True,
                      length=L)
    assert float((y - ref).abs().max()) < 1e-10


def test_module_tree_names_and_shapes():
    """resunet_with_multistft.py:40-118: ModuleDict branches keyed by str(win), FUSED_CH = 96 into encoder_block2,
    32 + 96 into decoder_block6's ConvBlockRes; FiLM names carry the ModuleDict key."""
    specs = {n: s for n, s, _ in arch.ms_param_specs()}
    for w in WINS:
        assert specs[f"base.pre_convs.{w}.weight"] == (32, 1, 1, 1)
        assert specs[f"base.encoder_block1s.{w}.conv_block1.conv1.weight"] == (32, 32, 3, 3)
        assert specs[f"film.encoder_block1s->{w}->conv_block1->beta2.weight"] == (32, 512)
    assert specs["base.bn0.weight"] == (1025,)
    assert specs["base.encoder_block2.conv_block1.conv1.weight"] == (64, 96, 3, 3)
    assert specs["base.encoder_block2.conv_block1.shortcut.weight"] == (64, 96, 1, 1)
    assert specs["base.decoder_block6.conv1.weight"] == (64, 32, 2, 2)
    assert specs["base.decoder_block6.conv_block2.conv1.weight"] == (32, 128, 3, 3)
    assert specs["film.decoder_block6->conv_block2->beta1.weight"] == (128, 512)
    assert specs["base.decoder_block5.conv_block2.conv1.weight"] == (64, 128, 3, 3)  # trunk unchanged
    from lass_amd.resunet_with_multistft import ResUNet30
    m = ResUNet30(1, 1, 512)
    assert set(m.state_dict()) == set(specs)
    assert m.film_meta["encoder_block1s"]["2048"]["conv_block1"] == {"beta1": 32, "beta2": 32}
    assert m.film_meta["decoder_block6"]["conv_block2"]["beta1"] == 128
    with pytest.raises(NotImplementedError):
        ResUNet30(1, 1, 512, win_lengths=(256, 2048))  # no 512 window to re-synthesise from


def test_oracle_forward_tiny_is_finite_and_branch_sensitive():
    sd = orr.to_torch(synthetic.make_state_dict_ms())
    _, mix = synthetic.make_mixtures(1, 8000)
    cond = torch.from_numpy(synthetic.make_condition(1))
    taps = {}
    out = oms.forward(sd, {"mixture": torch.from_numpy(mix)[:, None], "condition": cond}, taps=taps)["waveform"]
    assert out.shape == (1, 1, 8000) and torch.isfinite(out).all() and float(out.abs().max()) > 1e-4
    assert taps["x1"].shape == (1, 96, 64, 1024) and taps["x1_pool"].shape == (1, 96, 32, 512)
    # every analysis branch reaches the output
    sd2 = dict(sd)
    sd2["base.pre_convs.256.weight"] = sd["base.pre_convs.256.weight"] * 1.5
    out2 = oms.forward(sd2, {"mixture": torch.from_numpy(mix)[:, None], "condition": cond})["waveform"]
    assert float((out2 - out).abs().max()) > 1e-6


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_multistft_mirror_fails_loudly_on_cpu():
    from lass_amd._lib import LassError
    from lass_amd.resunet_with_multistft import ResUNet30
    m = ResUNet30()
    with pytest.raises(LassError):
        m({"mixture": torch.zeros(1, 1, 8000), "condition": torch.zeros(1, 512)})


# ---- GPU -----------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ms_sd():
    return synthetic.make_state_dict_ms()


@pytest.fixture(scope="module")
def ms_model(ms_sd):
    from lass_amd.resunet_with_multistft import ResUNet30
    m = ResUNet30(1, 1, 512)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ms_sd.items()})
    return m.to(DEV).eval()


def _rms(a):
    return float(a.double().pow(2).mean().sqrt())


@pytest.mark.gpu
@pytest.mark.parametrize("L", [16000, 8077, 1025])
def test_stft_components_common_nfft_vs_oracle(ms_model, L):
    """lass_stft_components (n_fft 2048, windows 256 / 512 / 2048, one launch) against the oracle's zero-padded-window
    STFT + magphase."""
    g = torch.Generator().manual_seed(L)
    x = (torch.rand(3, L, generator=g) * 2 - 1) * 0.5
    out = ms_model.engine.stft_components(x.to(DEV), 2048, WINS)
    for w in WINS:
        mag, cos, sin = (t.cpu() for t in out[w])
        m_ref, c_ref, s_ref = oms.stft_components(x.double(), w)
        assert mag.shape == m_ref.shape == (3, 1, 1 + L // 160, 1025)
        scale = float(m_ref.max())
        assert float((mag.double() - m_ref).abs().max()) < 2e-6 * scale + 1e-6
        assert float((mag.double() * cos.double() - m_ref * c_ref).abs().max()) < 4e-6 * scale + 1e-6
        assert float((mag.double() * sin.double() - m_ref * s_ref).abs().max()) < 4e-6 * scale + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("w", [512, 2048, 256])
def test_istft_nfft_vs_oracle_and_roundtrip(ms_model, w):
    L = 24000
    _, mix = synthetic.make_mixtures(2, L)
    x = torch.from_numpy(mix)
    re, im = oms.stft(x, w)
    eng = ms_model.engine
    y = eng.istft_nfft(re[:, 0].contiguous().to(DEV), im[:, 0].contiguous().to(DEV), L, 2048, w).cpu()
    ref = oms.istft(re, im, L, w)
    assert float((y - ref).abs().max()) < 2e-6
    assert float((y - x).abs().max()) < 2e-6   # hop 160 <= w / 1.6: perfect reconstruction


@pytest.mark.gpu
def test_multistft_tiny_vs_oracle_both_input_forms_and_taps(ms_model, ms_sd):
    sd = orr.to_torch(ms_sd)
    B, L = 2, 16000
    _, mix = synthetic.make_mixtures(B, L)
    cond = synthetic.make_condition(B)
    taps = {}
    ref = oms.forward(sd, {"mixture": torch.from_numpy(mix)[:, None], "condition": torch.from_numpy(cond)}, taps=taps)["waveform"]
    inp = {"mixture": torch.from_numpy(mix)[:, None].to(DEV), "condition": torch.from_numpy(cond).to(DEV)}
    out = ms_model(inp)["waveform"]
    assert out.shape == (B, 1, L)
    err = _rms(out.cpu() - ref)
    assert err <= 1e-5, err
    # taps read in place from the workspace: per-branch network inputs, the channel-concatenated skip and pool
    eng = ms_model.engine
    T = arch.frames_for(L)
    for w in WINS:
        x0 = eng.workspace_tensor(f"x0.{w}", B, L)
        assert float((x0.cpu() - taps[f"x0.{w}"]).abs().max()) < 2e-5 * max(1.0, float(taps[f"x0.{w}"].abs().max()))
    for name, key in (("encoder_block1", "x1"), ("encoder_block1.pool", "x1_pool"), ("encoder_block2", "encoder_block2"),
                      ("decoder_block5", "decoder_block5")):
        t = eng.workspace_tensor(name, B, L).cpu()
        assert t.shape == taps[key].shape, (name, t.shape, taps[key].shape)
        assert _rms(t - taps[key]) < 5e-6 * max(1.0, _rms(taps[key])), name
    o_re = eng.workspace_tensor("out_real", B, L)[:, :, :T].cpu()
    assert float((o_re - taps["out_real"]).abs().max()) < 2e-5 * max(1.0, float(taps["out_real"].abs().max()))
    assert torch.all(eng.workspace_tensor("out_real", B, L)[..., 1024] == 0)   # Nyquist bin exactly zero
    # the reference wrapper's input form: precomputed {win: (B,1,T,1025)} dicts + target_waveform
    from lass_amd import precompute_stfts as ps
    comp = ps.multi_resolution_stfts(inp["mixture"], WINS, n_fft=2048)
    d = {"stft_mixture_mag": {w: comp[w][0] for w in WINS}, "stft_mixture_cos": {w: comp[w][1] for w in WINS},
         "stft_mixture_sin": {w: comp[w][2] for w in WINS}, "condition": inp["condition"]}
    out2 = ms_model(d, target_waveform=inp["mixture"])["waveform"]
    assert out2.shape == (B, L)
    assert float((out2 - out[:, 0]).abs().max()) < 1e-6
    # and from CPU-resident precomputed tensors made by the oracle (the wire format as a file would deliver it)
    mag, cos, sin = oms.components(torch.from_numpy(mix)[:, None])
    d3 = {"stft_mixture_mag": mag, "stft_mixture_cos": cos, "stft_mixture_sin": sin, "condition": torch.from_numpy(cond)}
    out3 = ms_model(d3, target_waveform=torch.from_numpy(mix))["waveform"]
    assert _rms(out3.cpu() - ref[:, 0]) <= 1e-5


@pytest.mark.gpu
def test_multistft_ragged_lengths_and_batch_invariance(ms_model, ms_sd):
    sd = orr.to_torch(ms_sd)
    for B, L in ((1, 8000 + 77), (3, 5000)):
        _, mix = synthetic.make_mixtures(B, L)
        cond = synthetic.make_condition(B)
        ref = oms.forward(sd, {"mixture": torch.from_numpy(mix)[:, None], "condition": torch.from_numpy(cond)})["waveform"]
        out = ms_model({"mixture": torch.from_numpy(mix)[:, None].to(DEV), "condition": torch.from_numpy(cond).to(DEV)})["waveform"]
        assert _rms(out.cpu() - ref) <= 1e-5
        one = ms_model({"mixture": torch.from_numpy(mix)[:1, None].to(DEV), "condition": torch.from_numpy(cond)[:1].to(DEV)})["waveform"]
        assert torch.equal(one[0], out[0])


@pytest.mark.gpu
def test_multistft_config5_30s_32khz_vs_oracle(ms_model, ms_sd):
    """BASELINE configs[4] clip: 30 s @ 32 kHz = 960 000 samples, T = 6001 -> 6016 frames x 1024 bins; decoder_block6's
    concat is 3.15 GB per clip (beyond 2^31: exercises the unsigned 32-bit addressing of the Winograd kernels)."""
    sd = orr.to_torch(ms_sd)
    L = 960000
    segs = [synthetic.make_mixtures(1, 160000, first=20 + i)[1][0] for i in range(6)]
    mix = np.concatenate(segs)[None, :L].astype(np.float32)
    cond = synthetic.make_condition(1)
    out = ms_model({"mixture": torch.from_numpy(mix)[:, None].to(DEV), "condition": torch.from_numpy(cond).to(DEV)})["waveform"]
    torch.cuda.synchronize()
    torch.set_num_threads(16)
    ref = oms.forward(sd, {"mixture": torch.from_numpy(mix)[:, None], "condition": torch.from_numpy(cond)})["waveform"]
    err, sig = _rms(out.cpu() - ref), _rms(ref)
    assert sig > 1e-3 and err <= 1e-5, (err, sig)
    # the bf16 compute modes at the same size (their kernels read the 3.15-GB concat through unsigned offsets too)
    try:
        for mode, bar in (("bf16x3", 1e-5), ("bf16", 5e-2 * sig)):
            ms_model.set_compute_dtype(mode)
            o2 = ms_model({"mixture": torch.from_numpy(mix)[:, None].to(DEV), "condition": torch.from_numpy(cond).to(DEV)})["waveform"]
            e2 = _rms(o2.cpu() - ref)
            assert 1e-8 < e2 <= bar, (mode, e2)
    finally:
        ms_model.set_compute_dtype("f32")


@pytest.mark.gpu
def test_multistft_limits_and_modes(ms_model):
    from lass_amd._lib import LassError
    eng = ms_model.engine
    with pytest.raises(LassError):
        eng.workspace_bytes(1, 8192 * 160)      # concat would reach 4 GiB per clip
    assert eng.workspace_bytes(1, 8100 * 160) > 0
    # the same limit in the bf16 modes (their kernels' offset arithmetic is unsigned too)
    try:
        ms_model.set_compute_dtype("bf16")
        eng = ms_model.engine
        with pytest.raises(LassError):
            eng.workspace_bytes(1, 8192 * 160)
        assert eng.workspace_bytes(1, 8100 * 160) > 0
    finally:
        ms_model.set_compute_dtype("f32")


@pytest.mark.gpu
@pytest.mark.parametrize("mode,bar", [("bf16x3", 1e-5), ("bf16", 5e-2)])
def test_multistft_bf16_modes_vs_oracle(ms_model, ms_sd, mode, bar):
    """The bf16-MFMA compute modes on the multi-STFT separator (f32 tensors between the blocks): the split-operand mode is
    held to the f32 bar (1e-5 RMS), plain bf16 to 5 % of the signal RMS; ragged length, B = 2 and B = 1."""
    sd = orr.to_torch(ms_sd)
    try:
        ms_model.set_compute_dtype(mode)
        for B, L in ((2, 16000), (1, 8000 + 77)):
            _, mix = synthetic.make_mixtures(B, L)
            cond = synthetic.make_condition(B)
            ref = oms.forward(sd, {"mixture": torch.from_numpy(mix)[:, None], "condition": torch.from_numpy(cond)})["waveform"]
            out = ms_model({"mixture": torch.from_numpy(mix)[:, None].to(DEV), "condition": torch.from_numpy(cond).to(DEV)})["waveform"]
            err = _rms(out.cpu() - ref)
            assert err <= (bar if mode == "bf16x3" else bar * _rms(ref)), (mode, B, L, err)
            assert err > 1e-8   # really another arithmetic than the f32 path (1.7e-8 against the oracle)
    finally:
        ms_model.set_compute_dtype("f32")


@pytest.mark.gpu
def test_multistft_c_abi_error_paths(ms_model):
    """lass_create_multistft / lass_separate_components / lass_stft_components / lass_istft_nfft misuse: negative code +
    message, never a crash or a silent fallback."""
    from ctypes import POINTER, byref, c_int, c_void_p
    from lass_amd import _lib
    lib = _lib.load()
    h = c_void_p()
    wins = (c_int * 3)(256, 512, 2048)
    assert lib.lass_create_multistft(byref(h), 0, 1024, 3, wins, 512) < 0          # n_fft must be 2048
    assert b"2048" in lib.lass_last_error(None)
    assert lib.lass_create_multistft(byref(h), 0, 2048, 3, wins, 1024) < 0         # mask window not analysed
    assert lib.lass_create_multistft(byref(h), 0, 2048, 3, (c_int * 3)(256, 256, 512), 512) < 0   # duplicate window
    assert lib.lass_create_multistft(byref(h), 0, 2048, 2, (c_int * 2)(300, 512), 512) < 0        # not a power of two
    assert lib.lass_create_multistft(byref(h), 0, 2048, 5, wins, 512) < 0
    eng = ms_model.engine
    B, L = 1, 8000
    T = arch.frames_for(L)
    mags = [torch.rand(B, T, 1025, device=DEV) for _ in WINS]
    cos = torch.rand(B, T, 1025, device=DEV)
    cond = torch.zeros(B, 512, device=DEV)
    with pytest.raises(_lib.LassError):
        eng.separate_components(mags[:2], cos, cos, cond, L)                       # a branch is missing
    with pytest.raises(_lib.LassError):
        eng.separate_components(mags, cos[:, :-1].contiguous(), cos, cond, L)      # frame count mismatch
    with pytest.raises(_lib.LassError):
        eng.separate_components(mags, cos, cos, cond, L + 160)                     # target length inconsistent with T
    out = eng.separate_components(mags, cos, cos, cond, L)
    assert out.shape == (B, L) and torch.isfinite(out).all()
    x = torch.zeros(1, 4000, device=DEV)
    with pytest.raises(_lib.LassError):
        eng.stft_components(x, 2048, [300])
    with pytest.raises(_lib.LassError):
        eng.stft_components(x, 512, [256])                                          # n_fft 1024 or 2048 only
    with pytest.raises(_lib.LassError):
        eng.stft_components(torch.zeros(1, 1000, device=DEV), 2048, [512])         # shorter than the reflect padding
    with pytest.raises(_lib.LassError):
        eng.istft_nfft(torch.zeros(1, 10, 1025, device=DEV), torch.zeros(1, 10, 1025, device=DEV), 100000, 2048, 512)


@pytest.mark.gpu
def test_multistft_chunk_inference_stitches_like_whole_clip(ms_model):
    """chunk_inference (resunet.py:655-714 control flow) on the multi-STFT model: windows are separated 4 per launch and
    stitched; in the interior of each window's kept region the result must track the whole-clip forward (the network's
    receptive field is shorter than the 1 s of context on each side only approximately: loose bound)."""
    L = 400000
    segs = [synthetic.make_mixtures(1, 160000, first=30 + i)[1][0] for i in range(3)]
    mix = np.concatenate(segs)[None, None, :L].astype(np.float32)
    cond = synthetic.make_condition(1)
    inp = {"mixture": torch.from_numpy(mix).to(DEV), "condition": torch.from_numpy(cond).to(DEV)}
    out = ms_model.chunk_inference(inp)
    assert out.shape == (1, L) and out.dtype == np.float64 and np.isfinite(out).all()
    whole = ms_model(inp)["waveform"][0].cpu().numpy().astype(np.float64)
    mid = slice(32000 + 16000, 128000 - 16000)     # inside the first window's kept centre, away from its seams
    err = np.sqrt(np.mean((out[:, mid] - whole[:, mid]) ** 2))
    assert err < 0.1 * np.sqrt(np.mean(whole[:, mid] ** 2)), err


@pytest.mark.gpu
def test_dcase_evaluator_drives_the_multistft_model(tmp_path, ms_model, ms_sd):
    """The evaluator interface (dcase_evaluator.py:49-122) with the multi-STFT separator as `pl_model.ss_model`: same
    loop, same metrics, against the oracle evaluator running the multi-STFT oracle forward."""
    import os
    from lass_amd.audiosep import AudioSep, PrecomputedQueryEncoder
    from lass_amd.evaluator import DCASEEvaluator
    from oracle import evaluator as oev
    n, L = 5, 24000
    csv_path = synthetic.write_validation_set(str(tmp_path), n_clips=n, length=L)
    qe = PrecomputedQueryEncoder()
    ev = DCASEEvaluator(sampling_rate=16000, eval_indexes=csv_path, audio_dir=os.path.join(str(tmp_path), "lass_validation"),
                        batch_size=2)
    sisdr, sdri, sdr = ev(AudioSep(ss_model=ms_model, query_encoder=qe))
    clips = [synthetic.make_clip(i, L) for i in range(n)]
    conds = qe.get_query_embed("text", [f"synthetic tone cluster {i % 4}" for i in range(n)]).numpy()
    (o_sisdr, o_sdri, o_sdr), rows = oev.evaluate(orr.to_torch(ms_sd), clips, conds, forward=oms.forward)
    np.testing.assert_allclose(ev.last_rows, rows, atol=0.01)
    assert abs(sdr - o_sdr) < 0.01 and abs(sdri - o_sdri) < 0.01 and abs(sisdr - o_sisdr) < 0.01
