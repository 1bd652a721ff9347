"""Multi-resolution STFT front end + precomputed-STFT wire format.

CPU part: the oracle's per-window STFT is pinned to torch.stft for the reference's window set (config
stft_win_lengths [256, 512, 2048], hop 160), and the file format / dataset reader round-trip is exercised with CPU
tensors.  GPU part: lass_multi_stft (one launch for all windows) against the oracle."""
import pytest
import torch

from lass_amd import precompute_stfts as ps
from lass_amd import synthetic
from oracle import stft as ost

WINS = [256, 512, 2048]


@pytest.mark.parametrize("n_fft", [256, 512, 1024, 2048])
def test_oracle_window_set_equals_torch_stft(n_fft):
    L = 8077
    x = torch.randn(2, L, dtype=torch.float64, generator=torch.Generator().manual_seed(n_fft))
    re, im = ost.stft_fft(x, n_fft, 160)
    ts = torch.stft(x, n_fft, 160, n_fft, torch.hann_window(n_fft, periodic=True, dtype=torch.float64), center=True,
                    pad_mode="reflect", normalized=False, onesided=True, return_complex=True).transpose(1, 2)
    assert re.shape == (2, 1, 1 + L // 160, n_fft // 2 + 1)
    assert float((ts.real - re[:, 0]).abs().max()) < 1e-10 and float((ts.imag - im[:, 0]).abs().max()) < 1e-10
    mag, c, s = ost.stft_components(x, n_fft, 160)
    assert float((mag * c - re).abs().max()) < 1e-12 and float((mag * s - im).abs().max()) < 1e-12


def _fake_items(n, T=11):
    g = torch.Generator().manual_seed(n)
    items = []
    for k in range(n):
        st = {src: {w: tuple(torch.randn(1, 1, T, w // 2 + 1, generator=g) for _ in range(3)) for w in WINS}
              for src in ("mixture", "segment")}
        items.append({"stfts": st, "target_waveform": torch.randn(1, 1600, generator=g), "text": f"caption {k}",
                      "mixture_component_texts": [f"caption {k}", "noise"],
                      "stft_common_params": {"hop_length": 160, "window": "hann", "center": True, "pad_mode": "reflect"},
                      "stft_win_lengths": list(WINS)})
    return items


def test_wire_format_roundtrip_and_dataset_index(tmp_path):
    a, b = _fake_items(3), _fake_items(2)
    assert ps.save_batch_precomputed_data(tmp_path, 0, a) == 3
    assert ps.save_batch_precomputed_data(tmp_path, 10, b) == 2   # numeric, not lexicographic, file order
    assert ps.save_batch_precomputed_data(tmp_path, 2, []) == 0   # empty batch: no file (precompute_stfts.py:82-85)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["batch_000000.pt", "batch_000010.pt"]
    raw = torch.load(tmp_path / "batch_000000.pt", weights_only=True)
    assert isinstance(raw, list) and set(raw[0]) == {"stfts", "target_waveform", "text", "mixture_component_texts",
                                                     "stft_common_params", "stft_win_lengths"}
    assert set(raw[0]["stfts"]) == {"mixture", "segment"} and list(raw[0]["stfts"]["mixture"]) == WINS
    ds = ps.PrecomputedSTFTDataset(str(tmp_path), expected_num_items=5)
    assert len(ds) == 5 and ds.shard_sizes == [3, 2]
    for idx, src in enumerate(a + b):
        it = ds[idx]
        assert it["text"] == src["text"] and it["mixture_component_texts"] == src["mixture_component_texts"]
        for w in WINS:
            for got, want in zip(it["stfts"]["mixture"][w], src["stfts"]["mixture"][w]):
                assert got.shape == (1, 1, 11, w // 2 + 1) and torch.equal(got, want)
        assert torch.equal(it["target_waveform"], src["target_waveform"])
    with pytest.raises(IndexError):
        ds[5]
    with pytest.raises(FileNotFoundError):
        ps.PrecomputedSTFTDataset(str(tmp_path / "missing"))


def test_front_end_refuses_cpu_and_unsupported_configs():
    from lass_amd._lib import LassError
    x = torch.zeros(1, 1, 4000)
    with pytest.raises(LassError):
        ps.calculate_stft_components(x, 512, 160, 512, "hann", True, "reflect")
    with pytest.raises(LassError):  # common n_fft > win_length is supported (multi-STFT model input) - on the GPU only
        ps.calculate_stft_components(x, 2048, 160, 512, "hann", True, "reflect")
    with pytest.raises(NotImplementedError):
        ps.calculate_stft_components(x, 512, 160, 1024, "hann", True, "reflect")
    with pytest.raises(NotImplementedError):
        ps.calculate_stft_components(x, 300, 160, 300, "hann", True, "reflect")
    with pytest.raises(NotImplementedError):
        ps.calculate_stft_components(x, 512, 160, 512, "hamming", True, "reflect")


# ---- GPU -----------------------------------------------------------------------------------------------------------
def _check_against_oracle(out, x, wins, hop=160):
    for w in wins:
        mag, cos, sin = (t.cpu() for t in out[w])
        m_ref, c_ref, s_ref = ost.stft_components(x.double(), w, hop)
        assert mag.shape == m_ref.shape == (x.shape[0], 1, 1 + x.shape[1] // hop, w // 2 + 1)
        scale = float(m_ref.max())
        assert float((mag.double() - m_ref).abs().max()) < 2e-6 * scale + 1e-6
        # phase is ill-conditioned where |X| ~ 0: compare the re-synthesised real / imaginary parts
        assert float((mag.double() * cos.double() - m_ref * c_ref).abs().max()) < 4e-6 * scale + 1e-6
        assert float((mag.double() * sin.double() - m_ref * s_ref).abs().max()) < 4e-6 * scale + 1e-6
        strong = m_ref > 1e-2 * scale
        assert float((cos.double() - c_ref)[strong].abs().max()) < 2e-4
        assert float((sin.double() - s_ref)[strong].abs().max()) < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("L", [16000, 8077, 1025])
def test_multi_stft_vs_oracle(L):
    g = torch.Generator().manual_seed(L)
    x = (torch.rand(3, L, generator=g) * 2 - 1) * 0.5
    if L >= 16000:
        x[1] = torch.from_numpy(synthetic.make_mixtures(1, L)[1][0])
    x[2, : L // 2] = 0.0  # silence: |X| = 0 -> mag = 0, cos = sin = 0 (clamp on |X|, not |X|^2)
    out = ps.multi_resolution_stfts(x.cuda()[:, None, :], WINS, 160)
    _check_against_oracle(out, x, WINS)
    if L >= 8077:
        w = 256
        T_sil = (L // 2 - w // 2) // 160 - 1
        mag, cos, sin = out[w]
        assert not mag[2, 0, :T_sil].any() and not cos[2, 0, :T_sil].any() and not sin[2, 0, :T_sil].any()
    # the single-window entry point keeps the reference signature and equals the fused launch bit for bit
    one = ps.calculate_stft_components(x.cuda()[:, None, :], 512, 160, 512, "hann", True, "reflect")
    for a, b in zip(one, out[512]):
        assert a.is_contiguous() and torch.equal(a, b)
    _check_against_oracle(ps.multi_resolution_stfts(x.cuda(), [1024, 2048, 256, 512], 160), x, [1024])


@pytest.mark.gpu
def test_precompute_pipeline_10s_batch(tmp_path):
    """BASELINE-size clips (B=16 x 10 s) through the producer: items -> files -> dataset; Parseval-type size-independent
    check per window: sum_k c_k |X_k|^2 over a frame equals N * sum_n (w_n x_n)^2, c_k = 1 at DC/Nyquist and 2 elsewhere."""
    src, mix = synthetic.make_mixtures(16, 160000)
    mixtures, segments = torch.from_numpy(mix).cuda()[:, None, :], torch.from_numpy(src).cuda()[:, None, :]
    items = ps.make_precomputed_items(mixtures, segments, [f"c{k}" for k in range(16)], [["a", "b"]] * 16, WINS)
    assert len(items) == 16 and items[3]["stfts"]["segment"][2048][0].shape == (1, 1, 1001, 1025)
    for w in WINS:
        mag = torch.cat([it["stfts"]["mixture"][w][0] for it in items]).double().cpu()[:, 0]
        fr = ost.frame(torch.from_numpy(mix).double(), w, 160) * ost.hann_periodic(w)
        ck = torch.full((w // 2 + 1,), 2.0, dtype=torch.float64)
        ck[0] = ck[-1] = 1.0
        lhs, rhs = (mag ** 2 * ck).sum(-1), w * (fr ** 2).sum(-1)
        assert float(((lhs - rhs).abs() / (rhs + 1e-6)).max()) < 1e-4
    assert ps.save_batch_precomputed_data(tmp_path, 0, items[:8]) == 8
    assert ps.save_batch_precomputed_data(tmp_path, 1, items[8:]) == 8
    ds = ps.PrecomputedSTFTDataset(str(tmp_path), expected_num_items=16)
    it = ds[11]
    assert it["text"] == "c11" and not it["target_waveform"].is_cuda
    assert torch.equal(it["stfts"]["mixture"][512][1], items[11]["stfts"]["mixture"][512][1].cpu())
