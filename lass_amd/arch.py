"""Static description of the ResUNet30 separator: block table, parameter names/shapes, FiLM sites.

This is the single source of truth the host mirror (`lass_amd.resunet`), the C-ABI weight upload, the
synthetic-weight generator and the tests all read.  Names are the reference's `state_dict` keys so that
reference checkpoints load unchanged.

Reference: /root/reference/models/resunet.py
  - STFT geometry            :271-276  (n_fft = win = 1024, hop 160, centre, reflect, hann)
  - block table              :304-427
  - ConvBlockRes params      :84-145
  - DecoderBlockRes1B params :201-238  (bn2 exists, is in the state_dict, and is never used :230,254)
  - FiLM naming ('a->b->beta1', pre-order) :21-57, get_film_meta :598-618
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

N_FFT = 1024
HOP = 160
N_BINS = N_FFT // 2 + 1          # 513
F_CROP = N_BINS - 1              # 512   (resunet.py:552)
T_DOWN = 32                      # resunet.py:282
K_MASK = 3                       # resunet.py:280
BN_EPS = 1e-5                    # nn.BatchNorm2d default, resunet.py:98-99
LEAKY = 0.01                     # resunet.py:159-160,255
PRE_CH = 32                      # resunet.py:306-313


@dataclass(frozen=True)
class EncSpec:
    name: str
    cin: int
    cout: int
    down: Tuple[int, int]


@dataclass(frozen=True)
class DecSpec:
    name: str
    cin: int
    cout: int
    up: Tuple[int, int]


ENCODERS: Tuple[EncSpec, ...] = (
    EncSpec("encoder_block1", 32, 32, (2, 2)),
    EncSpec("encoder_block2", 32, 64, (2, 2)),
    EncSpec("encoder_block3", 64, 128, (2, 2)),
    EncSpec("encoder_block4", 128, 256, (2, 2)),
    EncSpec("encoder_block5", 256, 384, (2, 2)),
    EncSpec("encoder_block6", 384, 384, (1, 2)),
    EncSpec("conv_block7a", 384, 384, (1, 1)),
)

DECODERS: Tuple[DecSpec, ...] = (
    DecSpec("decoder_block1", 384, 384, (1, 2)),
    DecSpec("decoder_block2", 384, 384, (2, 2)),
    DecSpec("decoder_block3", 384, 256, (2, 2)),
    DecSpec("decoder_block4", 256, 128, (2, 2)),
    DecSpec("decoder_block5", 128, 64, (2, 2)),
    DecSpec("decoder_block6", 64, 32, (2, 2)),
)


def frames_for(length: int) -> int:
    """Centred STFT frame count (torchlibrosa STFT with center=True)."""
    return 1 + length // HOP


def padded_frames(t: int) -> int:
    """resunet.py:543-548."""
    return ((t + T_DOWN - 1) // T_DOWN) * T_DOWN


def _bn(prefix: str, c: int) -> List[Tuple[str, Tuple[int, ...], str]]:
    return [
        (prefix + ".weight", (c,), "bn_weight"),
        (prefix + ".bias", (c,), "bn_bias"),
        (prefix + ".running_mean", (c,), "bn_mean"),
        (prefix + ".running_var", (c,), "bn_var"),
        (prefix + ".num_batches_tracked", (), "bn_nbt"),
    ]


def _conv_block_res(prefix: str, cin: int, cout: int):
    out = []
    out += _bn(prefix + ".bn1", cin)
    out += _bn(prefix + ".bn2", cout)
    out.append((prefix + ".conv1.weight", (cout, cin, 3, 3), "conv_w"))
    out.append((prefix + ".conv2.weight", (cout, cout, 3, 3), "conv_w"))
    if cin != cout:
        out.append((prefix + ".shortcut.weight", (cout, cin, 1, 1), "conv_w"))
        out.append((prefix + ".shortcut.bias", (cout,), "bias"))
    return out


def film_sites() -> List[Tuple[str, int, bool]]:
    """(film module name, channels, used) in the reference's pre-order (resunet.py:598-618, :21-49).

    `decoder_blockN->beta2` is created (DecoderBlockRes1B.bn2 exists) but its output is never read.
    """
    sites: List[Tuple[str, int, bool]] = []
    for e in ENCODERS:
        sites.append((f"{e.name}->conv_block1->beta1", e.cin, True))
        sites.append((f"{e.name}->conv_block1->beta2", e.cout, True))
    for d in DECODERS:
        sites.append((f"{d.name}->beta1", d.cin, True))
        sites.append((f"{d.name}->beta2", d.cin, False))
        sites.append((f"{d.name}->conv_block2->beta1", 2 * d.cout, True))
        sites.append((f"{d.name}->conv_block2->beta2", d.cout, True))
    return sites


def param_specs(input_channels: int = 1, output_channels: int = 1, condition_size: int = 512):
    """Ordered [(state_dict key, shape, kind)] for `ResUNet30(input_channels, output_channels, condition_size)`.

    Keys carry the `base.` / `film.` prefixes of resunet.py:625-637.  torchlibrosa's frozen conv buffers
    (`base.stft.*`, `base.istft.*`) are deliberately absent: they are constants of the STFT geometry, not weights.
    """
    specs: List[Tuple[str, Tuple[int, ...], str]] = []
    specs += _bn("base.bn0", N_BINS)
    specs.append(("base.pre_conv.weight", (PRE_CH, input_channels, 1, 1), "conv_w"))
    specs.append(("base.pre_conv.bias", (PRE_CH,), "bias"))
    for e in ENCODERS:
        specs += _conv_block_res(f"base.{e.name}.conv_block1", e.cin, e.cout)
    for d in DECODERS:
        specs.append((f"base.{d.name}.conv1.weight", (d.cin, d.cout, d.up[0], d.up[1]), "tconv_w"))
        specs += _bn(f"base.{d.name}.bn1", d.cin)
        specs += _conv_block_res(f"base.{d.name}.conv_block2", 2 * d.cout, d.cout)
        specs += _bn(f"base.{d.name}.bn2", d.cin)
    specs.append(("base.after_conv.weight", (output_channels * K_MASK, PRE_CH, 1, 1), "conv_w"))
    specs.append(("base.after_conv.bias", (output_channels * K_MASK,), "bias"))
    for name, c, _used in film_sites():
        specs.append((f"film.{name}.weight", (c, condition_size), "linear_w"))
        specs.append((f"film.{name}.bias", (c,), "linear_b"))
    return specs


def conv_layer_table(t_pad: int, f: int = F_CROP) -> List[Dict]:
    """Every conv / transposed-conv the U-Net executes, with MAC counts per clip.

    Used by bench.py for the algorithmic-FLOP numerator and by DESIGN.md.
    """
    rows: List[Dict] = []
    h, w = t_pad, f
    rows.append(dict(name="pre_conv", kind="1x1", cin=1, cout=PRE_CH, h=h, w=w, macs=h * w * PRE_CH))
    skips = []
    for e in ENCODERS:
        rows.append(dict(name=e.name + ".conv1", kind="3x3", cin=e.cin, cout=e.cout, h=h, w=w,
                         macs=h * w * 9 * e.cin * e.cout))
        rows.append(dict(name=e.name + ".conv2", kind="3x3", cin=e.cout, cout=e.cout, h=h, w=w,
                         macs=h * w * 9 * e.cout * e.cout))
        if e.cin != e.cout:
            rows.append(dict(name=e.name + ".shortcut", kind="1x1", cin=e.cin, cout=e.cout, h=h, w=w,
                             macs=h * w * e.cin * e.cout))
        skips.append((h, w))
        h, w = h // e.down[0], w // e.down[1]
    skips.pop()  # conv_block7a's un-pooled output is discarded (resunet.py:562)
    for d in DECODERS:
        rows.append(dict(name=d.name + ".up", kind="tconv", cin=d.cin, cout=d.cout, h=h, w=w,
                         macs=h * w * d.cin * d.cout * d.up[0] * d.up[1]))
        h, w = h * d.up[0], w * d.up[1]
        assert (h, w) == skips.pop()
        rows.append(dict(name=d.name + ".conv1", kind="3x3", cin=2 * d.cout, cout=d.cout, h=h, w=w,
                         macs=h * w * 9 * 2 * d.cout * d.cout))
        rows.append(dict(name=d.name + ".conv2", kind="3x3", cin=d.cout, cout=d.cout, h=h, w=w,
                         macs=h * w * 9 * d.cout * d.cout))
        rows.append(dict(name=d.name + ".shortcut", kind="1x1", cin=2 * d.cout, cout=d.cout, h=h, w=w,
                         macs=h * w * 2 * d.cout * d.cout))
    rows.append(dict(name="after_conv", kind="1x1", cin=PRE_CH, cout=K_MASK, h=h, w=w, macs=h * w * PRE_CH * K_MASK))
    return rows


def wino4_routed(rows: List[Dict], min_cin: int = 32) -> set:
    """Names of the 3x3 rows of `conv_layer_table` that the f32 path runs as Winograd F(4x4,3x3) (csrc/wino4.hip; 36 instead
    of 144 MFMA multiplies per 4x4 output tile and (cin, cout) pair) - a mirror of the dispatch in csrc/api.hip
    (run_resblock) and lass_wino4_supported: at least `min_cin` input channels (LASS_WINO4, default 32; 0 = none), images
    whose width is a multiple of 32 and that tile into 8 x 64 or 16 x 32 pixel blocks; conv1 of every block, conv2 of the
    blocks with a 1x1 shortcut (decoder_block6's with the fused output head) and of encoder_block1 (residual = pre_conv(x0)).
    The identity blocks at 16 / 8 bins (encoder_block6, conv_block7a) and decoder_block1/2's narrow levels stay F(2x2,3x3)."""
    if min_cin <= 0:
        return set()
    with_shortcut = {r["name"].rsplit(".", 1)[0] for r in rows if r["name"].endswith(".shortcut")}
    out = set()
    for r in rows:
        if r["kind"] != "3x3":
            continue
        block, conv = r["name"].rsplit(".", 1)
        geom = r["w"] % 32 == 0 and ((r["w"] % 64 == 0 and r["h"] % 8 == 0) or r["h"] % 16 == 0)
        if not (geom and r["cin"] >= min_cin and r["cin"] % 8 == 0 and r["cout"] % 32 == 0):
            continue
        if conv == "conv1":
            out.add(r["name"])
        if conv == "conv2" and (block in with_shortcut or block.startswith("encoder_block1")):
            out.add(r["name"])
    return out


def conv_macs_per_clip(length: int) -> int:
    return sum(r["macs"] for r in conv_layer_table(padded_frames(frames_for(length))))


def conv3x3_bytes_per_clip(length: int, bytes_per_elem: int = 4) -> int:
    """Compulsory HBM bytes of the 26 conv3x3-class launches per clip with ideal per-launch fusion (every launch reads
    its input(s) once and writes its output(s) once; f32 storage): conv1 reads the block input, writes the activated
    intermediate; conv2 reads it, re-reads the block input for the residual / 1x1 shortcut, writes the block output and
    (encoders 1-6) the pooled output.  encoder_block1 reads the 1-channel x0 instead of a materialised pre_conv."""
    tp = padded_frames(frames_for(length))
    h, w = tp, F_CROP
    elems = 0
    for i, e in enumerate(ENCODERS):
        hw = h * w
        cin_read = 1 if i == 0 else e.cin
        elems += cin_read * hw + e.cout * hw            # conv1
        elems += e.cout * hw + cin_read * hw + e.cout * hw  # conv2 (+ residual / shortcut input)
        if i < 6:
            elems += e.cout * (h // e.down[0]) * (w // e.down[1])
        h, w = h // e.down[0], w // e.down[1]
    for d in DECODERS:
        h, w = h * d.up[0], w * d.up[1]
        hw = h * w
        elems += 2 * d.cout * hw + d.cout * hw            # conv1 over the concat
        elems += d.cout * hw + 2 * d.cout * hw + d.cout * hw  # conv2 + shortcut over the concat
    return elems * bytes_per_elem


# ---- multi-resolution-STFT separator (BASELINE configs[4]) ----------------------------------------
# Reference intent: /root/reference/models/resunet_with_multistft.py:40-118 (ctor), :137-216 (forward).  That file does
# not run as shipped (missing `.film` / `Dummy*` imports; per-window spectra of 129/257/1025 bins cannot be concatenated
# on the channel axis; one BatchNorm2d(257) is applied to all of them; decoder_block6's ConvBlockRes is built for 64
# input channels but receives 32 + 96), so the numbers below are an AUTHORED, coherent reading of it:
#   * every window is analysed at a COMMON n_fft = 2048 (its periodic Hann window zero-padded, centred - what
#     `calculate_stft_components(n_fft=2048, win_length=w)` of scripts/precompute_stfts.py:19-58 computes), so every
#     branch has 1025 bins -> 1024 after the Nyquist crop, and ONE bn0 over 1025 bins serves all branches;
#   * one `pre_convs[w]` (1 -> 32) + `encoder_block1s[w]` (32 -> 32, (2,2)) per window; pools and skips are concatenated
#     on channels in win_lengths order (96); encoder_block2 takes 96 channels; decoder_block6's ConvBlockRes takes
#     32 + 96 = 128 channels; everything else is the ResUNet30 trunk;
#   * the mask (sigmoid / tanh / magphase, resunet.py:469-495) is applied to the MASK_WINDOW (512) branch's
#     magnitude / phase and the waveform comes from an iSTFT at n_fft 2048 with that branch's window.
MS_N_FFT = 2048
MS_WIN_LENGTHS: Tuple[int, ...] = (256, 512, 2048)  # config stft_win_lengths (scripts/precompute_stfts.py:573-590)
MS_MASK_WINDOW = 512                                # resunet_with_multistft.py:185-188
MS_N_BINS = MS_N_FFT // 2 + 1                       # 1025
MS_F_CROP = MS_N_BINS - 1                           # 1024


def ms_encoders(win_lengths=MS_WIN_LENGTHS) -> Tuple[EncSpec, ...]:
    """Shared trunk encoders of the multi-STFT model (encoder_block2 widened to the fused channel count)."""
    fused = PRE_CH * len(win_lengths)
    return (EncSpec("encoder_block2", fused, 64, (2, 2)),) + ENCODERS[2:]


def ms_film_sites(win_lengths=MS_WIN_LENGTHS) -> List[Tuple[str, int, bool]]:
    """(film module name, channels, used) - get_film_meta's naming over the module tree of the multi-STFT base:
    ModuleDict children appear as 'encoder_block1s-><win>->conv_block1->beta1'."""
    fused = PRE_CH * len(win_lengths)
    sites: List[Tuple[str, int, bool]] = []
    for w in win_lengths:
        sites.append((f"encoder_block1s->{w}->conv_block1->beta1", PRE_CH, True))
        sites.append((f"encoder_block1s->{w}->conv_block1->beta2", PRE_CH, True))
    for e in ms_encoders(win_lengths):
        sites.append((f"{e.name}->conv_block1->beta1", e.cin, True))
        sites.append((f"{e.name}->conv_block1->beta2", e.cout, True))
    for d in DECODERS:
        cat = 2 * d.cout if d.name != "decoder_block6" else d.cout + fused
        sites.append((f"{d.name}->beta1", d.cin, True))
        sites.append((f"{d.name}->beta2", d.cin, False))
        sites.append((f"{d.name}->conv_block2->beta1", cat, True))
        sites.append((f"{d.name}->conv_block2->beta2", d.cout, True))
    return sites


def ms_param_specs(input_channels: int = 1, output_channels: int = 1, condition_size: int = 512,
                   win_lengths=MS_WIN_LENGTHS):
    """Ordered [(state_dict key, shape, kind)] of the multi-STFT `ResUNet30` (module names of
    resunet_with_multistft.py:40-118: `base.pre_convs.<w>`, `base.encoder_block1s.<w>.conv_block1`, ...)."""
    fused = PRE_CH * len(win_lengths)
    specs: List[Tuple[str, Tuple[int, ...], str]] = []
    specs += _bn("base.bn0", MS_N_BINS)
    for w in win_lengths:
        specs.append((f"base.pre_convs.{w}.weight", (PRE_CH, input_channels, 1, 1), "conv_w"))
        specs.append((f"base.pre_convs.{w}.bias", (PRE_CH,), "bias"))
    for w in win_lengths:
        specs += _conv_block_res(f"base.encoder_block1s.{w}.conv_block1", PRE_CH, PRE_CH)
    for e in ms_encoders(win_lengths):
        specs += _conv_block_res(f"base.{e.name}.conv_block1", e.cin, e.cout)
    for d in DECODERS:
        cat = 2 * d.cout if d.name != "decoder_block6" else d.cout + fused
        specs.append((f"base.{d.name}.conv1.weight", (d.cin, d.cout, d.up[0], d.up[1]), "tconv_w"))
        specs += _bn(f"base.{d.name}.bn1", d.cin)
        specs += _conv_block_res(f"base.{d.name}.conv_block2", cat, d.cout)
        specs += _bn(f"base.{d.name}.bn2", d.cin)
    specs.append(("base.after_conv.weight", (output_channels * K_MASK, PRE_CH, 1, 1), "conv_w"))
    specs.append(("base.after_conv.bias", (output_channels * K_MASK,), "bias"))
    for name, c, _used in ms_film_sites(win_lengths):
        specs.append((f"film.{name}.weight", (c, condition_size), "linear_w"))
        specs.append((f"film.{name}.bias", (c,), "linear_b"))
    return specs


def ms_conv_layer_table(t_pad: int, win_lengths=MS_WIN_LENGTHS) -> List[Dict]:
    """Convs of the multi-STFT model with MAC counts per clip (same row format as conv_layer_table)."""
    fused = PRE_CH * len(win_lengths)
    rows: List[Dict] = []
    h, w = t_pad, MS_F_CROP
    for wl in win_lengths:
        rows.append(dict(name=f"pre_convs.{wl}", kind="1x1", cin=1, cout=PRE_CH, h=h, w=w, macs=h * w * PRE_CH))
        for k in (1, 2):
            rows.append(dict(name=f"encoder_block1s.{wl}.conv{k}", kind="3x3", cin=PRE_CH, cout=PRE_CH, h=h, w=w,
                             macs=h * w * 9 * PRE_CH * PRE_CH))
    skips = [(h, w)]
    h, w = h // 2, w // 2
    for e in ms_encoders(win_lengths):
        rows.append(dict(name=e.name + ".conv1", kind="3x3", cin=e.cin, cout=e.cout, h=h, w=w, macs=h * w * 9 * e.cin * e.cout))
        rows.append(dict(name=e.name + ".conv2", kind="3x3", cin=e.cout, cout=e.cout, h=h, w=w, macs=h * w * 9 * e.cout * e.cout))
        if e.cin != e.cout:
            rows.append(dict(name=e.name + ".shortcut", kind="1x1", cin=e.cin, cout=e.cout, h=h, w=w, macs=h * w * e.cin * e.cout))
        skips.append((h, w))
        h, w = h // e.down[0], w // e.down[1]
    skips.pop()
    for d in DECODERS:
        cat = 2 * d.cout if d.name != "decoder_block6" else d.cout + fused
        rows.append(dict(name=d.name + ".up", kind="tconv", cin=d.cin, cout=d.cout, h=h, w=w,
                         macs=h * w * d.cin * d.cout * d.up[0] * d.up[1]))
        h, w = h * d.up[0], w * d.up[1]
        assert (h, w) == skips.pop()
        rows.append(dict(name=d.name + ".conv1", kind="3x3", cin=cat, cout=d.cout, h=h, w=w, macs=h * w * 9 * cat * d.cout))
        rows.append(dict(name=d.name + ".conv2", kind="3x3", cin=d.cout, cout=d.cout, h=h, w=w, macs=h * w * 9 * d.cout * d.cout))
        rows.append(dict(name=d.name + ".shortcut", kind="1x1", cin=cat, cout=d.cout, h=h, w=w, macs=h * w * cat * d.cout))
    rows.append(dict(name="after_conv", kind="1x1", cin=PRE_CH, cout=K_MASK, h=h, w=w, macs=h * w * PRE_CH * K_MASK))
    return rows
