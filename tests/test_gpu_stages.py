"""Stage-level GPU parity: the fused stages of the HIP path (avg-pool in conv2's epilogue, bn0/pad/crop in the STFT
epilogue, virtual concat) held directly to the oracle AND to the taps the reference itself produced (fixture G1,
tests/golden/g1_tiny.npz, made by tools/gen_golden.py from /root/reference/models/resunet.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lass_amd import arch, synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _relerr(got, ref):
    return float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-30))


@pytest.fixture(scope="module")
def oracle_sd(synthetic_sd):
    from oracle import resunet as orr
    return orr.to_torch(synthetic_sd)


@pytest.fixture(scope="module")
def model(synthetic_sd):
    from lass_amd.resunet import ResUNet30
    m = ResUNet30(1, 1, 512)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic_sd.items()})
    return m.to(DEV).eval()


ENC_CASES = [  # (encoder index, H, W): fused 2x2 pool, fused 1x2 pool (encoder_block6), odd H (stand-alone pool kernel)
    (0, 16, 64), (1, 24, 32), (2, 12, 64), (4, 8, 32), (5, 12, 16), (5, 6, 32), (1, 9, 32), (6, 4, 8)]


@pytest.mark.parametrize("ei,H,W", ENC_CASES)
def test_encoder_block_and_fused_pool_vs_oracle(model, oracle_sd, ei, H, W):
    """EncoderBlockRes1B.forward (resunet.py:186-198): block output AND F.avg_pool2d output, per block."""
    from oracle import resunet as orr
    e = arch.ENCODERS[ei]
    eng = model.engine
    g = torch.Generator().manual_seed(ei * 7919 + H * 100 + W)
    B = 2
    x = torch.randn(B, e.cin, H, W, generator=g)
    cond = torch.from_numpy(synthetic.make_condition(B))
    shift = eng.film(cond.to(DEV))
    y, pool = eng.encoder_block("base." + e.name, x.to(DEV), shift, e.cout, e.down)
    stem = f"{e.name}->conv_block1"
    ref = orr.conv_block_res(oracle_sd, f"base.{e.name}.conv_block1", x, orr.film(oracle_sd, cond, stem + "->beta1"),
                             orr.film(oracle_sd, cond, stem + "->beta2"))
    assert _relerr(y.cpu(), ref) < 5e-6
    if e.down == (1, 1):
        assert pool is None
        return
    ref_pool = F.avg_pool2d(ref, kernel_size=e.down)
    assert pool.shape == ref_pool.shape
    assert _relerr(pool.cpu(), ref_pool) < 5e-6
    # the pooled tensor must be the mean of the block output the same launch stored (fusion consistency, tight)
    assert float((pool - F.avg_pool2d(y, kernel_size=e.down)).abs().max()) < 2e-6 * max(1.0, float(y.abs().max()))


@pytest.mark.parametrize("L", [16000, 8000 + 77, 160000])
def test_front_end_x0_vs_oracle(model, oracle_sd, L):
    """resunet.py:533-552: bn0 over frequency, zero padding of T AFTER bn0 (not bn0(0)), Nyquist bin dropped."""
    from oracle import resunet as orr
    B = 2
    _, mix = synthetic.make_mixtures(B, L)
    taps = {}
    if L <= 16000:
        orr.forward(oracle_sd, {"mixture": torch.from_numpy(mix)[:, None],
                                "condition": torch.from_numpy(synthetic.make_condition(B))}, taps)
    mag, cos, sin, x0 = model.engine.front_end(torch.from_numpy(mix).to(DEV))
    T, Tp = arch.frames_for(L), arch.padded_frames(arch.frames_for(L))
    assert x0.shape == (B, Tp, 512)
    assert torch.all(x0[:, T:, :] == 0)  # padded frames are exact zeros
    if L <= 16000:
        ref = taps["x0"][:, 0]
        assert float((x0.cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    else:  # full size: x0 must be the affine bn0 image of the mag the same launch wrote
        sd = oracle_sd
        s = sd["base.bn0.weight"] / torch.sqrt(sd["base.bn0.running_var"] + 1e-5)
        h = sd["base.bn0.bias"] - sd["base.bn0.running_mean"] * s
        ref = mag.cpu()[:, :, :512] * s[:512] + h[:512]
        assert float((x0.cpu()[:, :T] - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))


def _sample(t, n=4096):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].cpu().numpy().astype(np.float32)


def _stats(t):
    d = t.detach().double()
    return np.asarray([d.sum().item(), d.abs().sum().item(), (d * d).sum().item(), float(d.numel())])


G1_TAPS = ["encoder_block1", "encoder_block1.pool", "encoder_block2.pool", "encoder_block3", "encoder_block4.pool",
           "encoder_block5.pool", "encoder_block6", "encoder_block6.pool", "conv_block7a", "decoder_block1.up",
           "decoder_block1", "decoder_block3.up", "decoder_block4", "decoder_block4.up", "decoder_block5",
           "decoder_block6.up", "out_real", "out_imag"]


def test_hip_taps_vs_reference_fixture_g1(model, golden_dir):
    """The HIP path's own intermediates (read in place from the workspace after lass_separate) against samples and
    moments of the tensors /root/reference/models/resunet.py produced for the same seeded inputs (fixture G1).  Covers
    the fused avg-pool stores, the virtual concat halves, every transposed conv and the fused output head."""
    g = np.load(os.path.join(golden_dir, "g1_tiny.npz"))
    B, L = 2, 16000
    _, mix = synthetic.make_mixtures(B, L)
    cond = synthetic.make_condition(B)
    eng = model.engine
    out = eng.separate(torch.from_numpy(mix).to(DEV), torch.from_numpy(cond).to(DEV))
    torch.cuda.synchronize()
    assert float((out.cpu() - torch.from_numpy(g["waveform"][:, 0])).pow(2).mean().sqrt()) < 2e-6
    T = arch.frames_for(L)
    for name in G1_TAPS:
        t = eng.workspace_tensor(name, B, L).clone()
        if name in ("out_real", "out_imag"):
            t = t[:, :, :T]  # (B,1,T,513) as the reference holds it
        scale = max(1.0, float(np.sqrt(g[name + "/stats"][2] / g[name + "/stats"][3])))
        st = _stats(t)
        assert st[3] == g[name + "/stats"][3], name
        np.testing.assert_allclose(_sample(t), g[name + "/sample"], rtol=0, atol=3e-4 * scale, err_msg=name)
        np.testing.assert_allclose(st[:3], g[name + "/stats"][:3], rtol=2e-5, atol=1e-2, err_msg=name)
    xc = eng.workspace_tensor("conv_block7a", B, L)
    np.testing.assert_allclose(xc.cpu().numpy(), g["x_center"], rtol=0, atol=2e-4)


def test_hip_taps_vs_oracle_every_element(model, oracle_sd):
    """Every element of every intermediate the shipped f32 pipeline leaves in its workspace (all 7 skips, 6 pooled tensors,
    the bottleneck, 6 transposed-conv halves, decoder blocks 1-5, the separated spectrum) against the oracle's tensor of the
    same name, at L = 25 600 (161 -> 192 frames: F(4x4,3x3) levels, F(2x2,3x3) levels and the 6 x 8-bin bottom): the maximum
    absolute deviation of each tensor must stay below 3e-5 of that tensor's largest magnitude - an isolated wrong pixel (a
    tile edge, a halo rule, a fused-pool window) fails it; the samples-and-moments fixtures G1 / G4 hold the same tensors
    to the reference's own values."""
    from oracle import resunet as orr
    B, L = 2, 25600
    _, mix = synthetic.make_mixtures(B, L, first=8)
    cond = synthetic.make_condition(B, seed=5)
    eng = model.engine
    eng.separate(torch.from_numpy(mix).to(DEV), torch.from_numpy(cond).to(DEV))
    torch.cuda.synchronize()
    taps = {}
    orr.forward(oracle_sd, {"mixture": torch.from_numpy(mix)[:, None, :], "condition": torch.from_numpy(cond)}, taps=taps)
    T = arch.frames_for(L)
    names = ([f"encoder_block{i}" for i in range(1, 7)] + [f"encoder_block{i}.pool" for i in range(1, 7)] + ["conv_block7a"] +
             [f"decoder_block{i}.up" for i in range(1, 7)] + [f"decoder_block{i}" for i in range(1, 6)] + ["out_real", "out_imag"])
    worst = (0.0, None)
    for name in names:
        t = eng.workspace_tensor(name, B, L).clone().cpu()
        ref = taps[name]
        if name in ("out_real", "out_imag"):
            t = t[:, :, :T]
        if t.shape != ref.shape:   # the oracle keeps the frame padding / a channel axis the workspace tensor does not
            ref = ref.reshape(t.shape) if ref.numel() == t.numel() else ref[..., :t.shape[-2], :t.shape[-1]]
        assert t.shape == ref.shape, (name, t.shape, ref.shape)
        rel = float((t - ref).abs().max()) / float(ref.abs().max())
        worst = max(worst, (rel, name))
        assert rel < 3e-5, (name, rel)
    print("largest max-abs deviation / max-abs over the", len(names), "tensors:", worst)


UPS_REST = [("decoder_block3", 384, 256, (2, 2), 6, 32), ("decoder_block4", 256, 128, (2, 2), 10, 64),
            ("decoder_block2", 384, 384, (2, 2), 3, 16)]


@pytest.mark.parametrize("name,cin,cout,up,h,w", UPS_REST)
def test_upconv_remaining_decoders_vs_oracle(model, oracle_sd, name, cin, cout, up, h, w):
    """The transposed convs not covered by test_gpu_parity.py::test_upconv_vs_oracle (decoders 3, 4; odd h on 2)."""
    from oracle import resunet as orr
    eng = model.engine
    g = torch.Generator().manual_seed(h * 100 + w)
    B = 2
    x = torch.randn(B, cin, h, w, generator=g)
    cond = torch.from_numpy(synthetic.make_condition(B))
    shift = eng.film(cond.to(DEV))
    y = eng.upconv("base." + name, x.to(DEV), shift, cout, up).cpu()
    hh = F.leaky_relu(orr._bn(oracle_sd, f"base.{name}.bn1", x) + orr.film(oracle_sd, cond, f"{name}->beta1"), 0.01)
    ref = F.conv_transpose2d(hh, oracle_sd[f"base.{name}.conv1.weight"], stride=up)
    assert y.shape == ref.shape
    assert _relerr(y, ref) < 2e-6


def test_workspace_tensor_rejects_unknown_names(model):
    from lass_amd._lib import LassError
    eng = model.engine
    with pytest.raises(LassError):
        eng.workspace_tensor("no_such_tensor", 2, 16000)


def test_separate_rejects_bad_out_and_overlong_clips(model):
    """`out` is validated before its pointer crosses the C-ABI; clips beyond the 32-bit per-clip addressing
    limit are refused with a message instead of silently reading zeros."""
    from lass_amd._lib import LassError
    eng = model.engine
    mix = torch.zeros(2, 16000, device=DEV)
    cond = torch.zeros(2, 512, device=DEV)
    with pytest.raises(LassError):
        eng.separate(mix, cond, out=torch.empty(2, 8000, device=DEV))
    with pytest.raises(LassError):
        eng.separate(mix, cond, out=torch.empty(2, 32000, device=DEV)[:, ::2])
    with pytest.raises(LassError):
        eng.separate(mix, cond, out=torch.empty(2, 16000, device=DEV, dtype=torch.float64))
    # decoder_block6's concat: 131 072 B per padded frame, below 4 GiB (f32 Winograd kernels) -> 32 736 frames
    with pytest.raises(LassError):
        eng.workspace_bytes(1, 32736 * 160)
    assert eng.workspace_bytes(1, 32736 * 160 - 1) > 0


def test_graph_replay_equals_eager(synthetic_sd, monkeypatch):
    """lass_separate captures its launches into a hipGraph when the same (pointers, shape) call repeats and replays it
    afterwards: outputs must be bit-identical to the eager path, also after the INPUT CONTENT behind the same pointers
    changes, and fresh buffers must fall back to eager without a re-capture per call."""
    from lass_amd.resunet import ResUNet30

    def make():
        m = ResUNet30(1, 1, 512)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic_sd.items()})
        return m.to(DEV).eval()

    B, L = 2, 24000
    _, mix = synthetic.make_mixtures(2 * B, L)
    cond = torch.from_numpy(synthetic.make_condition(B)).to(DEV)
    x = torch.from_numpy(mix[:B]).to(DEV)
    eng = make().engine
    out = torch.empty_like(x)
    outs = []
    for i in range(5):
        eng.separate(x, cond, out)
        outs.append(out.clone())
    on, caps, reps = eng.graph_stats()
    assert on and caps == 1 and reps == 2, (on, caps, reps)   # eager, eager, capture, replay, replay
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    x.copy_(torch.from_numpy(mix[B:]).to(DEV))                # same pointers, new audio: the replay must see it
    eng.separate(x, cond, out)
    replay_new = out.clone()
    assert eng.graph_stats()[2] == 3 and not torch.equal(replay_new, outs[0])
    monkeypatch.setenv("LASS_GRAPH", "0")
    eng0 = make().engine
    assert eng0.graph_stats()[0] is False
    ref_new = eng0.separate(x, cond)
    assert torch.equal(ref_new, replay_new)
    ref_old = eng0.separate(torch.from_numpy(mix[:B]).to(DEV), cond)
    assert torch.equal(ref_old, outs[0])
    monkeypatch.delenv("LASS_GRAPH")
    # fresh output buffers every call (what the evaluator does): no capture churn
    caps_before = eng.graph_stats()[1]
    for _ in range(3):
        y = eng.separate(torch.from_numpy(mix[:B]).to(DEV).clone(), cond)
    assert torch.equal(y, outs[0]) and eng.graph_stats()[1] <= caps_before + 1
    # lass_set_graph_replay (round 4, bench.py's eager leg): off -> the same call launches eagerly (no replay counted), same
    # bits; on again -> the cached graph is replayed without a new capture.  An argument error while a capture would be due
    # is reported as such and leaves replay enabled (round-3 advice).
    caps0, reps0 = eng.graph_stats()[1:]
    eng.set_graph_replay(False)
    assert eng.graph_stats()[0] is False
    eng.separate(x, cond, out)
    assert eng.graph_stats()[1:] == (caps0, reps0) and torch.equal(out, replay_new)
    eng.set_graph_replay(True)
    eng.separate(x, cond, out)
    assert eng.graph_stats() == (True, caps0, reps0 + 1) and torch.equal(out, replay_new)


def test_workspace_reuse_and_graphs_across_ragged_batches(synthetic_sd):
    """The evaluator's pattern (dcase_evaluator.py:65-116 batched): the common batch B=16... and a ragged tail.  The smaller
    shape must reuse the ONE workspace allocation (lass_separate only needs workspace_bytes >= its plan), so the workspace
    pointer - part of the graph key - is stable, both shapes keep their captured graph (<= 2 captures however often they
    alternate), and results equal fresh eager runs."""
    from lass_amd.resunet import ResUNet30
    m = ResUNet30(1, 1, 512)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic_sd.items()})
    eng = m.to(DEV).eval().engine
    L = 24000
    _, mix = synthetic.make_mixtures(16, L)
    xs = {16: torch.from_numpy(mix).to(DEV), 5: torch.from_numpy(mix[:5]).to(DEV)}
    cs = {b: torch.from_numpy(synthetic.make_condition(b)).to(DEV) for b in xs}
    os_ = {b: torch.empty_like(xs[b]) for b in xs}
    first = {}
    for rnd in range(4):
        for b in (16, 5):
            eng.separate(xs[b], cs[b], os_[b])
            if rnd == 0:
                first[b] = os_[b].clone()
            else:
                assert torch.equal(os_[b], first[b]), (rnd, b)
    on, caps, reps = eng.graph_stats()
    assert eng.ws_allocations == 1
    assert on and caps == 2 and reps == 2, (on, caps, reps)   # per shape: eager, eager, capture, replay
    assert torch.equal(first[5], first[16][:5])                # clips are independent of their batch
    # a larger shape grows the buffer once; the small ones keep working in it
    big = eng.separate(torch.from_numpy(synthetic.make_mixtures(2, 2 * 16 * L // 2)[1]).to(DEV),
                       torch.from_numpy(synthetic.make_condition(2)).to(DEV))
    assert torch.isfinite(big).all() and eng.ws_allocations <= 2
    assert torch.equal(eng.separate(xs[5], cs[5]), first[5])


@pytest.mark.parametrize("B,H,W", [(2, 40, 64), (3, 18, 32), (1, 64, 96)])
def test_wino32_resident_kernel_vs_oracle_and_wino(synthetic_sd, oracle_sd, monkeypatch, B, H, W):
    """wino32.hip (weights-resident persistent kernel of the 32-cout full-resolution layers, resunet.py:147-165 at the
    shapes of :315-323,408-418) against the oracle and against wino.hip (LASS_WINO32=0), block by block: encoder_block1 with
    its fused avg-pool (CONV1_ACT / CONV2_IDENT), decoder_block6's ConvBlockRes (Cin = 64 conv1, conv2 + transform-domain
    shortcut), encoder_block2 (32 -> 64 as two 32-cout slices: conv1, conv2 with Cin = 64 + shortcut + pool), encoder_block3 (conv1
    64 -> 128 as four slices).  Shapes cover border-only images (one block column), strips that do not fill a block of 8 (H/2 = 9, 20),
    and B = 3."""
    from lass_amd.engine import Engine
    from oracle import resunet as orr
    g = torch.Generator().manual_seed(H * W + B)
    cond = torch.from_numpy(synthetic.make_condition(B))
    outs = {}
    monkeypatch.setenv("LASS_WINO4", "0")  # (round 4: these layers run as F(4x4,3x3) by default; wino32.hip stays the F(2x2,3x3) form)
    for sw in ("1", "0"):
        monkeypatch.setenv("LASS_WINO32", sw)
        e = Engine(DEV)
        e.load_state_dict(synthetic_sd)
        monkeypatch.delenv("LASS_WINO32")
        shift = e.film(cond.to(DEV))
        x1 = torch.randn(B, 32, H, W, generator=g) if sw == "1" else x1
        x6 = torch.randn(B, 64, H, W, generator=g) if sw == "1" else x6
        y1, p1 = e.encoder_block("base.encoder_block1", x1.to(DEV), shift, 32, (2, 2))
        y6 = e.convblock("base.decoder_block6.conv_block2", x6.to(DEV), shift, 32)
        y2, p2 = e.encoder_block("base.encoder_block2", x1.to(DEV), shift, 64, (2, 2))   # two 32-cout slices
        y3, _ = e.encoder_block("base.encoder_block3", x6.to(DEV), shift, 128, (2, 2))  # conv1 as four slices
        outs[sw] = (y1.cpu(), p1.cpu(), y6.cpu(), y2.cpu(), p2.cpu(), y3.cpu())
    r1 = orr.conv_block_res(oracle_sd, "base.encoder_block1.conv_block1", x1,
                            orr.film(oracle_sd, cond, "encoder_block1->conv_block1->beta1"),
                            orr.film(oracle_sd, cond, "encoder_block1->conv_block1->beta2"))
    r6 = orr.conv_block_res(oracle_sd, "base.decoder_block6.conv_block2", x6,
                            orr.film(oracle_sd, cond, "decoder_block6->conv_block2->beta1"),
                            orr.film(oracle_sd, cond, "decoder_block6->conv_block2->beta2"))
    r2 = orr.conv_block_res(oracle_sd, "base.encoder_block2.conv_block1", x1,
                            orr.film(oracle_sd, cond, "encoder_block2->conv_block1->beta1"),
                            orr.film(oracle_sd, cond, "encoder_block2->conv_block1->beta2"))
    r3 = orr.conv_block_res(oracle_sd, "base.encoder_block3.conv_block1", x6,
                            orr.film(oracle_sd, cond, "encoder_block3->conv_block1->beta1"),
                            orr.film(oracle_sd, cond, "encoder_block3->conv_block1->beta2"))
    refs = (r1, torch.nn.functional.avg_pool2d(r1, (2, 2)), r6, r2, torch.nn.functional.avg_pool2d(r2, (2, 2)), r3)
    for got, old, ref in zip(outs["1"], outs["0"], refs):
        assert got.shape == ref.shape
        assert _relerr(got, ref) < 5e-6, _relerr(got, ref)     # the bar of test_convblock_vs_oracle
        assert _relerr(got, old) < 2e-6, _relerr(got, old)     # same products, another summation order in B^T d B


W4_BLOCKS = [  # (module prefix, film site stem, cin, cout, H, W): shapes with one block column / row, image edges on every side
    ("base.decoder_block5.conv_block2", "decoder_block5->conv_block2", 128, 64, 16, 64),    # 8 x 64 blocks, two block rows
    ("base.decoder_block4.conv_block2", "decoder_block4->conv_block2", 256, 128, 32, 32),   # 16 x 32 blocks (W % 64 != 0)
    ("base.decoder_block3.conv_block2", "decoder_block3->conv_block2", 512, 256, 8, 128),   # one block row, two block columns
    ("base.decoder_block2.conv_block2", "decoder_block2->conv_block2", 768, 384, 16, 32),
    ("base.encoder_block4.conv_block1", "encoder_block4->conv_block1", 128, 256, 24, 64),
]


@pytest.mark.parametrize("prefix,stem,cin,cout,H,W", W4_BLOCKS)
def test_wino4_convblock_vs_oracle_and_wino(synthetic_sd, oracle_sd, monkeypatch, prefix, stem, cin, cout, H, W):
    """wino4.hip - conv1 of a ConvBlockRes (resunet.py:101-119,147-165) as Winograd F(4x4,3x3): 36 instead of 64 MFMA multiplies
    per 16 outputs, 6x6 transforms (constants up to 8 and down to 1/24) - against the oracle and against the F(2x2,3x3) kernels
    (LASS_WINO4=0) on the same block.  Rounding error per layer is ~6x that of F(2x2,3x3): the block bar is 2e-5 relative here
    (5e-6 for the F(2x2,3x3) path, test_convblock_vs_oracle); the end-to-end bar stays north_star's 1e-4 RMS."""
    from lass_amd.engine import Engine
    from oracle import resunet as orr
    B = 2
    g = torch.Generator().manual_seed(H * 1000 + W + cin)
    x = torch.randn(B, cin, H, W, generator=g)
    cond = torch.from_numpy(synthetic.make_condition(B))
    ys = {}
    for sw in ("64", "0"):
        monkeypatch.setenv("LASS_WINO4", sw)
        e = Engine(DEV)
        e.load_state_dict(synthetic_sd)
        monkeypatch.delenv("LASS_WINO4")
        ys[sw] = e.convblock(prefix, x.to(DEV), e.film(cond.to(DEV)), cout).cpu()
    ref = orr.conv_block_res(oracle_sd, prefix, x, orr.film(oracle_sd, cond, stem + "->beta1"),
                             orr.film(oracle_sd, cond, stem + "->beta2"))
    e4, e2 = _relerr(ys["64"], ref), _relerr(ys["0"], ref)
    print(prefix, "relative RMS error vs oracle: F(4x4,3x3)", e4, " F(2x2,3x3)", e2, " max abs", float((ys["64"] - ref).abs().max()))
    assert e2 < 5e-6 and e4 < 2e-5, (e4, e2)
    assert float((ys["64"] - ref).abs().max()) < 3e-4 * max(1.0, float(ref.abs().max()))
    assert not torch.equal(ys["64"], ys["0"])   # really another kernel


def test_wino4_encoder_block_with_fused_pool(synthetic_sd, oracle_sd, monkeypatch):
    """conv2 + direct 1x1 shortcut + bias + the block's fused 2x2 avg-pool in wino4.hip (resunet.py:147-165,186-198):
    encoder_block4 (128 -> 256) and encoder_block3 (64 -> 128) as whole blocks against the oracle."""
    from lass_amd.engine import Engine
    from oracle import resunet as orr
    e = Engine(DEV)   # default: LASS_WINO4 = 32
    e.load_state_dict(synthetic_sd)
    B = 2
    cond = torch.from_numpy(synthetic.make_condition(B))
    shift = e.film(cond.to(DEV))
    g = torch.Generator().manual_seed(77)
    for name, cin, cout, H, W in (("encoder_block4", 128, 256, 16, 64), ("encoder_block3", 64, 128, 32, 32),
                                  ("encoder_block1", 32, 32, 24, 128), ("encoder_block2", 32, 64, 16, 32)):
        x = torch.randn(B, cin, H, W, generator=g)
        y, pool = e.encoder_block("base." + name, x.to(DEV), shift, cout, (2, 2))
        ref = orr.conv_block_res(oracle_sd, f"base.{name}.conv_block1", x, orr.film(oracle_sd, cond, f"{name}->conv_block1->beta1"),
                                 orr.film(oracle_sd, cond, f"{name}->conv_block1->beta2"))
        assert _relerr(y.cpu(), ref) < 2e-5, (name, _relerr(y.cpu(), ref))
        assert _relerr(pool.cpu(), F.avg_pool2d(ref, (2, 2))) < 2e-5, name


def test_hip_path_vs_reference_fixture_g4(golden_dir):
    """Fixture G4 (tools/gen_golden.py --only g4): the reference's own models/resunet.py output for a SECOND seeded weight set,
    other conditions and clips, L = 48 000 (301 -> 320 frames: every level from 320 x 512 down to 10 x 16, F(4x4,3x3) at the
    levels that tile).  The HIP path is held to it on the full waveform, on decoder_block1's output in full and on row /
    column margins of five large taps and of the separated spectrum - every element counted, so an isolated wrong pixel (a
    tile edge, a padding rule) cannot hide behind a strided sample (round 3's weak spot)."""
    from lass_amd.resunet import ResUNet30
    from test_oracle_golden import G4_SEED, G4_TAPS, check_against_g4
    g = np.load(os.path.join(golden_dir, "g4_second_weights.npz"))
    B, L = 2, 48000
    m = ResUNet30(1, 1, 512)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synthetic.make_state_dict(seed=G4_SEED).items()})
    m = m.to(DEV).eval()
    _, mix = synthetic.make_mixtures(B, L, first=20)
    cond = synthetic.make_condition(B, seed=G4_SEED)
    out = m({"mixture": torch.from_numpy(mix)[:, None, :].to(DEV), "condition": torch.from_numpy(cond).to(DEV)})["waveform"]
    torch.cuda.synchronize()
    T = arch.frames_for(L)
    taps = {}
    for n in G4_TAPS + ("decoder_block1",):
        t = m.engine.workspace_tensor(n, B, L).clone()
        if n in ("out_real", "out_imag"):
            t = t[:, :, :T]
        taps[n] = t.cpu().numpy()
    check_against_g4(g, out.cpu().numpy(), taps)
    # the bf16 modes against the same reference output: split operands inside the f32 tolerance class, plain bf16 within 5 %
    ref = torch.from_numpy(g["waveform"])
    inp = {"mixture": torch.from_numpy(mix)[:, None, :].to(DEV), "condition": torch.from_numpy(cond).to(DEV)}
    for mode, bound in (("bf16x3", 1e-5), ("bf16", 5e-2 * float(ref.pow(2).mean().sqrt()))):
        w = m.set_compute_dtype(mode)(inp)["waveform"].cpu()
        err = float((w - ref).pow(2).mean().sqrt())
        print("G4", mode, "waveform RMS error vs the reference's own output", err)
        assert err <= bound, (mode, err)
