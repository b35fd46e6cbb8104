"""The six-term arithmetic pinned at KERNEL level (round-3 verdict, item 7).  fp32 products formed from exact three-way bf16
splits (hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate) are
driven through ONE layer at a time (cm_debug_conv_io: no GroupNorm / SiLU / time row / residual, bias kept) with hostile
operands and compared (a) with an fp64 host evaluation of the same convolution and (b) with the SAME layer on the fp32 matrix
instruction (mode 1).  Layers: a full-resolution and a half-resolution Winograd layer (conv_wino_p_kernel), a quarter-resolution
whole-sample layer (conv_qr2_kernel) and an upsample conv (conv_ups_kernel) of the ATC model
(/root/reference/models/backbones/layers.py:30-43,84,93-96 as wired at unet.py:45-115).

Stated bounds (error per output relative to S = sum |x| |w| + |bias|, the backward-error scale of a dot product):
  * wide dynamic range (log-uniform 1e-30 .. 1e30, random signs) and heavy cancellation (common offset 1e6, zero-mean weights):
    e6 <= 4 e32 + 1e-7 and e6 <= 4e-6, where e32 is the fp32-instruction form's error on the same data;
  * inputs around and below the fp32 normal range (1e-44 .. 1e-36): absolute error <= 1e-6 S + 2e-38 (the low-order bf16 terms of
    such values are subnormal and may be flushed; nothing is amplified);
  * NaN / Inf: the set of NON-FINITE outputs is identical to the fp32 form's (an Inf input becomes NaN in the six-term form --
    its remainder Inf - Inf -- where the fp32 instruction yields +-Inf: same positions, documented);
  * |x| > 3.3962e38 (the midpoint between the largest finite bf16, 3.3895e38, and 2^128): the hi term rounds to Inf, so the
    six-term form returns non-finite values at exactly the outputs whose receptive field holds such an element, where fp32
    instructions still return finite numbers -- the one documented deviation from fp32 arithmetic, the top 0.2 % of the fp32
    range; up to 3.396e38 the split is exact like everywhere else."""
import ctypes as C
import zlib

import numpy as np
import pytest

from crowdmod_ddpm_4d_amd import native, spec
from helpers import SEED_W, full_cfg, synth_inputs

pytestmark = pytest.mark.gpu

LAYERS = {   # label: kind
    "encoder_blocks.0.conv_1.weight": "conv",            # 32 -> 32, 8 x 12 x 36: full-resolution six-term Winograd
    "encoder_blocks.2.conv_2.weight": "conv",            # 64 -> 64, 4 x 6 x 18: two-tile Winograd
    "bottleneck_blocks.1.conv_1.weight": "conv",         # 128 -> 128, 2 x 3 x 9: conv_qr2
    "decoder_blocks.5.upsample.1.weight": "ups",         # 64 -> 64 onto the full resolution: conv_ups
}
B = 2


@pytest.fixture(scope="module")
def net():
    from crowdmod_ddpm_4d_amd.unet import UNet
    cfg = full_cfg(3)
    n = UNet(input_channels=3, output_channels=3, num_res_blocks=1, base_channels=32, base_channels_multiples=(1, 2, 4),
             apply_attention=(False, False, True), dropout_rate=0.1, time_multiple=4, condition="Past", max_batch=B)
    params = spec.init_params(cfg, SEED_W)
    n.load_state_dict(params)
    past, fut = synth_inputs(B, 3, 12, 36, 5, 3, "hostile")
    n(fut, np.array([5, 900]), past)          # tiles are tuned at the first launch
    n._params_for_test = params
    return n


def _find(net, label):
    L, h = native.lib(), net._handle
    cnt = C.c_int32()
    native.check(L.cm_debug_conv_count(h, C.byref(cnt)))
    buf = C.create_string_buffer(512)
    for i in range(cnt.value):
        native.check(L.cm_debug_conv_info(h, i, buf, len(buf)))
        f = buf.value.decode().split()
        if f[0] == "conv" and f[1] == label:
            return i, dict(Ci=int(f[5]), Co=int(f[6]), Zo=int(f[7]), Yo=int(f[8]), Xo=int(f[9]))
    raise KeyError(label)


def _run(net, idx, mode, x, out_shape):
    out = np.empty(out_shape, dtype=np.float32)
    native.check(native.lib().cm_debug_conv_io(net._handle, idx, mode, x.ctypes.data, None, out.ctypes.data, x.shape[0]))
    return out


def _ref64(x, w, bias, ups):
    """fp64 evaluation in the internal layout: x [B][Z][Y][X][Ci]; w the state_dict tensor [Co][Ci][kH][kW][kL]
    (tap (dz, dy, dx) = element [dy][dx][dz]); returns (y, S) with S = sum |x| |w| + |bias|."""
    x = x.astype(np.float64)
    if ups:
        x = x.repeat(2, axis=1).repeat(2, axis=2).repeat(2, axis=3)
    Bn, Z, Y, X, Ci = x.shape
    w64 = w.astype(np.float64)
    y = np.zeros((Bn, Z, Y, X, w.shape[0])) + bias.astype(np.float64)
    S = np.zeros_like(y) + np.abs(bias.astype(np.float64))
    xp = np.zeros((Bn, Z + 2, Y + 2, X + 2, Ci))
    xp[:, 1:-1, 1:-1, 1:-1] = x
    ap = np.abs(xp)
    with np.errstate(invalid="ignore", over="ignore"):
        for dz in range(3):
            for dy in range(3):
                for dx in range(3):
                    wt = w64[:, :, dy, dx, dz].T                      # [Ci][Co]
                    sl = (slice(None), slice(dz, dz + Z), slice(dy, dy + Y), slice(dx, dx + X))
                    y += xp[sl] @ wt
                    S += ap[sl] @ np.abs(wt)
    return y, S


def _inputs(kind, shape, seed):
    rng = np.random.default_rng(seed)
    if kind == "wide":
        return (rng.choice([-1.0, 1.0], size=shape) * 10.0 ** rng.uniform(-30, 30, size=shape)).astype(np.float32)
    if kind == "cancel":
        return (1.0e6 + rng.standard_normal(shape)).astype(np.float32)
    if kind == "pairs":                                  # +- pairs along the channel axis at large magnitude
        a = (rng.standard_normal(shape) * 1.0e4).astype(np.float32)
        a[..., 1::2] = -a[..., 0::2]
        return a
    if kind == "tiny":
        return (rng.choice([-1.0, 1.0], size=shape) * 10.0 ** rng.uniform(-44, -36, size=shape)).astype(np.float32)
    raise KeyError(kind)


@pytest.mark.parametrize("label", list(LAYERS))
@pytest.mark.parametrize("kind", ["wide", "cancel", "pairs"])
def test_six_term_error_stays_within_a_multiple_of_the_fp32_instruction_error(net, label, kind):
    idx, g = _find(net, label)
    ups = LAYERS[label] == "ups"
    zs, ys, xs = (g["Zo"] // 2, g["Yo"] // 2, g["Xo"] // 2) if ups else (g["Zo"], g["Yo"], g["Xo"])
    x = _inputs(kind, (B, zs, ys, xs, g["Ci"]), zlib.crc32(f"{label}/{kind}".encode()) & 0xFFFF)
    w = np.asarray(net._params_for_test[label], dtype=np.float32)
    bias = np.asarray(net._params_for_test[label.replace(".weight", ".bias")], dtype=np.float32)
    oshape = (B, g["Zo"], g["Yo"], g["Xo"], g["Co"])
    y6 = _run(net, idx, 0, x, oshape)
    y32 = _run(net, idx, 1, x, oshape)
    assert not np.array_equal(y6, y32), "mode 1 must withhold the six-term fragments"
    ref, S = _ref64(x, w, bias, ups)
    assert np.isfinite(ref).all() and np.isfinite(y6).all() and np.isfinite(y32).all()
    e6 = float((np.abs(y6 - ref) / S).max())
    e32 = float((np.abs(y32 - ref) / S).max())
    print(f"{label} {kind}: e6 {e6:.3e} e32 {e32:.3e}")
    assert e6 <= 4.0 * e32 + 1e-7, (label, kind, e6, e32)
    assert e6 <= 4e-6, (label, kind, e6)


@pytest.mark.parametrize("label", list(LAYERS))
def test_six_term_on_values_at_the_bottom_of_the_fp32_range(net, label):
    idx, g = _find(net, label)
    ups = LAYERS[label] == "ups"
    zs, ys, xs = (g["Zo"] // 2, g["Yo"] // 2, g["Xo"] // 2) if ups else (g["Zo"], g["Yo"], g["Xo"])
    x = _inputs("tiny", (B, zs, ys, xs, g["Ci"]), 77)
    w = np.asarray(net._params_for_test[label], dtype=np.float32)
    bias = np.zeros(g["Co"], dtype=np.float32)
    y6 = _run(net, idx, 0, x, (B, g["Zo"], g["Yo"], g["Xo"], g["Co"]))
    ref, S = _ref64(x, w, np.asarray(net._params_for_test[label.replace(".weight", ".bias")], dtype=np.float32), ups)
    err = np.abs(y6 - ref)
    assert np.isfinite(y6).all()
    assert (err <= 1e-6 * S + 2e-38).all(), float((err - 1e-6 * S).max())
    del bias


@pytest.mark.parametrize("label", list(LAYERS))
def test_non_finite_inputs_poison_the_same_outputs_as_on_fp32_instructions(net, label):
    idx, g = _find(net, label)
    ups = LAYERS[label] == "ups"
    zs, ys, xs = (g["Zo"] // 2, g["Yo"] // 2, g["Xo"] // 2) if ups else (g["Zo"], g["Yo"], g["Xo"])
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, zs, ys, xs, g["Ci"])).astype(np.float32)
    x[0, 0, 0, 0, 1] = np.nan
    x[0, zs - 1, ys - 1, xs - 1, 2] = np.inf
    x[1, zs // 2, ys // 2, xs // 2, 0] = -np.inf
    oshape = (B, g["Zo"], g["Yo"], g["Xo"], g["Co"])
    y6 = _run(net, idx, 0, x, oshape)
    y32 = _run(net, idx, 1, x, oshape)
    bad6, bad32 = ~np.isfinite(y6), ~np.isfinite(y32)
    assert bad32.any() and np.array_equal(bad6, bad32), (int(bad6.sum()), int(bad32.sum()))
    ok = ~bad32
    assert np.abs(y6[ok] - y32[ok]).max() <= 1e-4
    # the rest of the batch and of the volume is untouched: receptive fields are local
    assert ok.mean() > 0.5


@pytest.mark.parametrize("label", ["encoder_blocks.0.conv_1.weight", "bottleneck_blocks.1.conv_1.weight"])
def test_documented_deviation_beyond_the_largest_finite_bf16(net, label):
    """|x| = 3.40e38 is a finite fp32 value whose bf16 rounding is Inf: the six-term form poisons the outputs that read it, the
    fp32 instruction does not.  Pinned so that the deviation stays exactly this narrow."""
    idx, g = _find(net, label)
    rng = np.random.default_rng(6)
    x = rng.standard_normal((B, g["Zo"], g["Yo"], g["Xo"], g["Ci"])).astype(np.float32)
    big = np.float32(3.40e38)
    assert np.isfinite(big)
    x[0, 1, 1, 1, 3] = big
    oshape = (B, g["Zo"], g["Yo"], g["Xo"], g["Co"])
    y6 = _run(net, idx, 0, x, oshape)
    y32 = _run(net, idx, 1, x, oshape)
    assert np.isfinite(y32).all()
    bad = ~np.isfinite(y6)
    assert bad.any() and not bad[1].any()
    zz, yy, xx = np.nonzero(bad[0].any(axis=-1))
    # only the neighbourhood of the element (the Winograd layer poisons whole 2 x 2 output patches: one more row / column)
    assert zz.max() <= 2 and yy.max() <= 3 and xx.max() <= 3
    x[0, 1, 1, 1, 3] = np.float32(3.395e38)                           # rounds DOWN to the largest finite bf16: exact again
    y6b = _run(net, idx, 0, x, oshape)
    assert np.isfinite(y6b).all()
