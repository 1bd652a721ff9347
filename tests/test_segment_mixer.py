"""SegmentMixer (mixer half): `lass_segment_mix` and its host mirror against the oracle's restatement of
data/waveform_mixers.py:19-92.  PARITY UNPINNED: the reference module imports `pyloudnorm` (absent) and ships no vectors, so the
restatement is held to the source text by closed forms (CPU part) and the HIP path to the restatement (GPU part)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import waveform_mixers as owm


def test_oracle_mixer_closed_forms():
    rng = np.random.default_rng(3)
    L = 4000
    a = torch.from_numpy(rng.standard_normal(L).astype(np.float32)) * 0.1
    b = torch.from_numpy(rng.standard_normal(L).astype(np.float32)) * 0.3
    x = torch.stack([a, b])
    # 0 dB everywhere: the neighbour is brought to the primary's energy, and so is the noise sum -> noise energy = primary energy
    mix, seg = owm.segment_mix(x, [2, 2], [[0.0], [0.0]], [0.0, 0.0])
    noise0 = mix[0] - seg[0]
    assert abs(float(owm.get_energy(noise0) / owm.get_energy(a)) - 1.0) < 1e-4
    assert torch.equal(seg[0], a) and float(mix.abs().max()) <= 1.0
    # +6 dB on the noise sum: energy ratio 10^(6/10)
    mix6, _ = owm.segment_mix(x, [2, 2], [[0.0], [0.0]], [6.0, 6.0])
    assert abs(float(owm.get_energy(mix6[0] - a) / owm.get_energy(a)) - 10 ** 0.6) < 1e-3
    # the ratio clamp: a neighbour 10^6 times stronger is divided by 50 only (waveform_mixers.py:80)
    big = torch.stack([a * 1e-3, b * 1e3])
    r = owm.get_energy_ratio(big[1], big[0])
    assert float(r) == 50.0
    # declipping: both outputs scaled by 0.9 / max when the mixture exceeds 1 (:49-53)
    loud = torch.stack([a * 8, b * 8])
    m, s = owm.segment_mix(loud, [2, 2], [[10.0], [10.0]], [10.0, 10.0])
    assert abs(float(m[0].abs().max()) - 0.9) < 1e-6 and float(s[0].abs().max()) < float(loud[0].abs().max())


def test_host_mirror_draws_in_the_reference_order():
    """waveform_mixers.py:34 (mix_num), :39 -> :88 (one dB draw per mixed-in clip), :43 -> :88 (the noise sum), per clip."""
    from lass_amd.waveform_mixers import SegmentMixer
    for max_mix in (2, 4):
        random.seed(1234)
        got = SegmentMixer(max_mix, -10, 10).draw(9)
        random.seed(1234)
        ref = owm.draws_like_reference(9, max_mix, -10, 10)
        for g, r in zip(got, ref):
            assert np.array_equal(g, r)
        random.seed(1234)   # and literally the reference's sequence of calls
        for n in range(9):
            mn = random.randint(2, max_mix)
            assert mn == got[0][n]
            for i in range(1, mn):
                assert random.randint(-10, 10) == got[1][n, i - 1]
            assert random.randint(-10, 10) == got[2][n]
    with pytest.raises(NotImplementedError):
        SegmentMixer(1, -10, 10)
    with pytest.raises(Exception, match="MI355X"):
        SegmentMixer(2, -10, 10)(torch.zeros(2, 1, 100))


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,max_mix", [(6, 32000, 2), (5, 16000 + 77, 4), (16, 160000, 2), (1, 8000, 2), (3, 12000, 8)])
def test_segment_mix_vs_oracle(B, L, max_mix):
    from lass_amd import synthetic
    from lass_amd.engine import get_engine
    from lass_amd.waveform_mixers import SegmentMixer
    rng = np.random.default_rng(B * 1000 + max_mix)
    clips = np.stack([synthetic.make_clip(i, L)[i % 2] for i in range(B)]).astype(np.float32)
    clips[0] *= 6.0         # a loud primary: its mixture clips -> the declip branch
    if B > 2:
        clips[2] *= 1e-4    # a faint clip: ratio clamps at 50 (as a neighbour) and at 0.02 (as the reference)
    if B > 4:
        clips[4] = 0.0      # silence: energy floor 1e-10
    # (B = 1: the reference's wrap-around index (n + i) % batch_size mixes the clip with itself; max_mix 8 > B: indices wrap twice)
    random.seed(99)
    draws = owm.draws_like_reference(B, max_mix, -10, 10)
    x = torch.from_numpy(clips)
    o_mix, o_seg = owm.segment_mix(x, *draws)
    eng = get_engine(torch.device("cuda:0"))
    mix, seg = eng.segment_mix(x.cuda(), torch.from_numpy(draws[0]), torch.from_numpy(draws[1]), torch.from_numpy(draws[2]))
    for b in range(B):
        scale = max(float(o_mix[b].abs().max()), 1e-3)
        assert float((mix[b].cpu() - o_mix[b]).abs().max()) < 2e-6 * max(scale, 1.0) + 2e-6 * scale, b
        assert float((seg[b].cpu() - o_seg[b]).abs().max()) < 2e-6 * max(scale, 1.0), b
    peaks = [float(m.abs().max()) for m in o_mix]
    assert any(abs(p - 0.9) < 1e-5 for p in peaks) and (B == 1 or any(p < 0.9 for p in peaks))   # both declip branches exercised
    # the host mirror: same draws from the same seed, (B, 1, L) in -> (B, 1, L) out, input untouched
    random.seed(99)
    xin = x.cuda()[:, None, :].clone()
    m2, s2 = SegmentMixer(max_mix, -10, 10)(xin)
    assert m2.shape == (B, 1, L) and torch.equal(m2[:, 0], mix) and torch.equal(s2[:, 0], seg) and torch.equal(xin[:, 0].cpu(), x)
    rc_err = None
    try:
        eng.segment_mix(x.cuda(), torch.from_numpy(draws[0]), torch.zeros(B, 9), torch.from_numpy(draws[2]))
    except Exception as e:  # max_mix_num - 1 > 7 is refused by the C-ABI
        rc_err = str(e)
    assert rc_err and "max_comp" in rc_err


@pytest.mark.gpu
def test_mixer_feeds_the_precompute_pipeline(tmp_path):
    """scripts/precompute_stfts.py:352-622: segments -> SegmentMixer -> multi-resolution STFTs -> per-item dicts -> shard."""
    from lass_amd import precompute_stfts as ps
    from lass_amd import synthetic
    from lass_amd.waveform_mixers import SegmentMixer
    from oracle import stft as ostft
    B, L = 4, 16000
    clips = np.stack([synthetic.make_clip(i, L)[0] for i in range(B)]).astype(np.float32)
    texts = [f"clip {i}" for i in range(B)]
    random.seed(5)
    items = ps.mix_and_make_precomputed_items(torch.from_numpy(clips).cuda()[:, None, :], texts, SegmentMixer(3, -10, 10),
                                              [256, 512, 2048])
    random.seed(5)
    draws = owm.draws_like_reference(B, 3, -10, 10)
    o_mix, o_seg = owm.segment_mix(torch.from_numpy(clips), *draws)
    assert len(items) == B
    for n, it in enumerate(items):
        assert it["text"] == texts[n]
        assert it["mixture_component_texts"] == [texts[n]] + [texts[(n + i) % B] for i in range(1, int(draws[0][n]))]
        assert float((it["target_waveform"].cpu().flatten() - o_seg[n]).abs().max()) < 2e-6
        mag = it["stfts"]["mixture"][512][0]
        assert mag.shape == (1, 1, 1 + L // 160, 257)
        ref_mag = ostft.stft_components(o_mix[n][None], 512, 160)[0]   # calculate_stft_components at n_fft = win_length = 512
        assert float((mag.cpu().flatten() - ref_mag.flatten().float()).abs().max()) < 2e-4 * max(1.0, float(ref_mag.abs().max()))
    assert ps.save_batch_precomputed_data(tmp_path, 0, items) == B
    assert len(ps.PrecomputedSTFTDataset(str(tmp_path))) == B
