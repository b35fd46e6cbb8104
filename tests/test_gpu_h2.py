"""The default plan's "h2" arithmetic pinned at KERNEL level: fp32 products formed from f16 TWO-way splits (x = hi + mid, 22 of
the 24 mantissa bits; cross terms hi*mid, mid*hi, hi*hi on v_mfma_f32_32x32x16_f16, fp32 accumulate) in
the layers whose input is GroupNorm + SiLU output -- the Winograd 3x3x3 layers (conv_wino_p_kernel), the quarter-resolution
layers (conv_qr2_kernel) and the last conv (conv_fin_kernel) of /root/reference/models/backbones/unet.py:45-122.  One layer at a
time (cm_debug_conv_io mode 2: raw sources, bias kept, h2 form allowed) on operands INSIDE the range the plan guarantees
(|x| <= 8000 = the static bound 32000 / the Winograd transform's gain 4; cm_model.cpp: h2_act_bounded), against an fp64 host
evaluation and against the same layer on the fp32 matrix instruction (mode 1) and in the six-term bf16 form (mode 0).

Stated bounds, per output, with S = sum |x| |w| + |bias| and W1 = sum |w| over the receptive field:
  * |e_h2| <= 4 e32 S + 1e-7 S + 2.4e-7 W1   (e32: the fp32 instruction's relative error on the same data; the last term is the
    ABSOLUTE floor of an f16 two-way split: below |x| = 0.125 the mid term is a subnormal f16 with spacing 2^-24, so an element
    carries up to 2^-25 = 3e-8 of absolute error -- times the Winograd transforms' gain; at the network's operating point, values
    O(1) and outputs O(1), that floor is 1e-7 of the output and the forward error equals the six-term form's, see
    tests/test_gpu_parity.py and profiles/round4_fwd_err.txt);
  * finite outputs for every input inside the bound; an element beyond 65504 / 4 is outside the contract (the plan never lets
    one reach these kernels: its static bound keeps the layer on the six-term bf16 form instead -- last test)."""
import ctypes as C
import zlib

import numpy as np
import pytest

from crowdmod_ddpm_4d_amd import native, spec
from helpers import SEED_W, full_cfg, synth_inputs
from test_gpu_six_term_hostile import _find, _ref64, _run

pytestmark = pytest.mark.gpu

LAYERS = [
    "encoder_blocks.0.conv_1.weight",            # 32 -> 32, 8 x 12 x 36: full-resolution Winograd
    "encoder_blocks.2.conv_2.weight",            # 64 -> 64, 4 x 6 x 18: two-tile Winograd
    "bottleneck_blocks.1.conv_1.weight",         # 128 -> 128, 2 x 3 x 9: conv_qr2
    "final.2.weight",                            # 32 -> 3, 8 x 12 x 36: conv_fin
]
B = 2


@pytest.fixture(scope="module")
def net():
    from crowdmod_ddpm_4d_amd.unet import UNet
    cfg = full_cfg(3)
    n = UNet(input_channels=3, output_channels=3, num_res_blocks=1, base_channels=32, base_channels_multiples=(1, 2, 4),
             apply_attention=(False, False, True), dropout_rate=0.1, time_multiple=4, condition="Past", max_batch=B)
    params = spec.init_params(cfg, SEED_W)
    n.load_state_dict(params)
    past, fut = synth_inputs(B, 3, 12, 36, 5, 3, "h2")
    n(fut, np.array([5, 900]), past)
    n._params_for_test = params
    return n


def _out_stride(net, idx):
    buf = C.create_string_buffer(512)
    native.check(native.lib().cm_debug_conv_info(net._handle, idx, buf, len(buf)))
    return int(buf.value.decode().split()[17])


def _inputs(kind, shape, seed):
    rng = np.random.default_rng(seed)
    sgn = rng.choice([-1.0, 1.0], size=shape)
    if kind == "unit":                                   # the operating point: SiLU of a standard normal
        v = rng.standard_normal(shape)
        return (v / (1.0 + np.exp(-v))).astype(np.float32)
    if kind == "range":                                  # nine decades inside the bound
        return (sgn * 10.0 ** rng.uniform(-6, 3, size=shape)).astype(np.float32)
    if kind == "edge":                                   # at the static bound
        return (sgn * rng.uniform(4000.0, 8000.0, size=shape)).astype(np.float32)
    if kind == "cancel":                                 # common offset, the signal in the low bits
        return (1.0e3 + rng.standard_normal(shape)).astype(np.float32)
    if kind == "tiny":                                   # below the f16 range altogether
        return (sgn * 10.0 ** rng.uniform(-12, -5, size=shape)).astype(np.float32)
    raise KeyError(kind)


def _w1(w, shape):
    """sum |w| over the receptive field of every output (zero padding included as zeros)."""
    ones = np.ones(shape, dtype=np.float32)
    y, _ = _ref64(ones, np.abs(w), np.zeros(w.shape[0], dtype=np.float32), False)
    return y


@pytest.mark.parametrize("label", LAYERS)
@pytest.mark.parametrize("kind", ["unit", "range", "edge", "cancel", "tiny"])
def test_h2_error_bound_on_bounded_operands(net, label, kind):
    idx, g = _find(net, label)
    shape = (B, g["Zo"], g["Yo"], g["Xo"], g["Ci"])
    x = _inputs(kind, shape, zlib.crc32(f"h2/{label}/{kind}".encode()) & 0xFFFF)
    w = np.asarray(net._params_for_test[label], dtype=np.float32)
    bias = np.asarray(net._params_for_test[label.replace(".weight", ".bias")], dtype=np.float32)
    oshape = (B, g["Zo"], g["Yo"], g["Xo"], _out_stride(net, idx))      # (the last conv's output tensor is padded to 4 channels)
    yh = _run(net, idx, 2, x, oshape)[..., :g["Co"]]
    y6 = _run(net, idx, 0, x, oshape)[..., :g["Co"]]
    y32 = _run(net, idx, 1, x, oshape)[..., :g["Co"]]
    ref, S = _ref64(x, w, bias, False)
    W1 = _w1(w, shape)
    assert np.isfinite(yh).all() and np.isfinite(ref).all()
    assert not np.array_equal(yh, y6) or kind == "tiny", "mode 2 must run the h2 form"
    eh, e6, e32 = np.abs(yh - ref), np.abs(y6 - ref), np.abs(y32 - ref)
    r32 = float((e32 / S).max())
    slack = eh - (4.0 * r32 + 1e-7) * S - 2.4e-7 * W1
    print(f"{label} {kind}: h2 {float((eh / S).max()):.3e} six {float((e6 / S).max()):.3e} fp32 {r32:.3e} of S; "
          f"floor use {float((eh / (2.4e-7 * W1)).max()):.2f}")
    assert float(slack.max()) <= 0.0, (label, kind, float(slack.max()))
    if kind in ("unit", "edge", "cancel"):               # where no element sits on the absolute floor: as good as the six-term form
        assert float((eh / S).max()) <= 2.0 * float((e6 / S).max()) + 1e-7, (label, kind)


@pytest.mark.parametrize("kind", ["unit", "range", "cancel", "huge", "minute", "wide"])
def test_h2_upsample_conv_takes_its_range_from_the_data(net, kind):
    """The upsample conv's source is RAW (a block output): its h2 form multiplies the staged values by a per-sample power of two
    derived from the source tensor's slot statistics (|x| <= |mean| + sqrt(M2) per channel; cm_h2_sample_scale), so ANY magnitude
    whose statistics are finite is in range -- 1e15-sized and 1e-30-sized tensors included -- and only elements far below their
    sample's maximum sit on the f16 floor, where they no longer matter to S = sum |x| |w|.  Bound: |e| <= (4 e32 + 2e-7) S.
    Beyond |x| ~ 1e17 the sum of squares behind M2 overflows fp32 -- as it does for GroupNorm itself -- and the h2 form returns
    non-finite values (loud, not wrong): second test below."""
    label = "decoder_blocks.5.upsample.1.weight"
    idx, g = _find(net, label)
    zs, ys, xs = g["Zo"] // 2, g["Yo"] // 2, g["Xo"] // 2
    shape = (B, zs, ys, xs, g["Ci"])
    rng = np.random.default_rng(zlib.crc32(f"h2ups/{kind}".encode()) & 0xFFFF)
    sgn = rng.choice([-1.0, 1.0], size=shape)
    if kind == "huge":
        x = (sgn * rng.uniform(1e14, 1e15, size=shape)).astype(np.float32)
    elif kind == "minute":
        x = (sgn * 10.0 ** rng.uniform(-32, -30, size=shape)).astype(np.float32)
    elif kind == "wide":
        x = (sgn * 10.0 ** rng.uniform(-30, 15, size=shape)).astype(np.float32)
    else:
        x = _inputs(kind, shape, 11)
    w = np.asarray(net._params_for_test[label], dtype=np.float32)
    bias = np.asarray(net._params_for_test[label.replace(".weight", ".bias")], dtype=np.float32)
    oshape = (B, g["Zo"], g["Yo"], g["Xo"], g["Co"])
    yh = _run(net, idx, 2, x, oshape)
    y6 = _run(net, idx, 0, x, oshape)
    y32 = _run(net, idx, 1, x, oshape)
    ref, S = _ref64(x, w, bias, True)
    assert np.isfinite(yh).all() and np.isfinite(ref).all()
    assert not np.array_equal(yh, y6) or kind == "minute", "mode 2 must run the h2 form"   # (minute: the bias is all that is left in fp32)
    eh, e6, r32 = float((np.abs(yh - ref) / S).max()), float((np.abs(y6 - ref) / S).max()), float((np.abs(y32 - ref) / S).max())
    print(f"upsample {kind}: h2 {eh:.3e} six {e6:.3e} fp32 {r32:.3e} of S")
    assert eh <= 4.0 * r32 + 2e-7, (kind, eh, r32)


def test_h2_upsample_conv_is_loud_when_the_statistics_overflow(net):
    """|x| ~ 1e19: M2 = sum (x - mean)^2 is Inf in fp32, the bound is Inf, no scale exists: every output of that sample is
    non-finite (the six-term bf16 form of mode 0 still returns finite numbers there -- the one regime it covers and h2 does
    not; a GroupNorm over such a tensor is Inf / NaN in the reference as well)."""
    label = "decoder_blocks.5.upsample.1.weight"
    idx, g = _find(net, label)
    shape = (B, g["Zo"] // 2, g["Yo"] // 2, g["Xo"] // 2, g["Ci"])
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape).astype(np.float32)
    x[0] *= np.float32(1e19)
    oshape = (B, g["Zo"], g["Yo"], g["Xo"], g["Co"])
    yh = _run(net, idx, 2, x, oshape)
    y6 = _run(net, idx, 0, x, oshape)
    assert np.isfinite(y6).all()
    assert not np.isfinite(yh[0]).any() and np.isfinite(yh[1]).all()      # per-sample: the other sample is untouched


def test_a_layer_whose_groupnorm_affine_breaks_the_static_bound_keeps_the_six_term_form():
    """gamma x 1000 on one GroupNorm: sqrt(n) max|gamma| exceeds the bound the f16 range allows, so that layer must stay on
    the bf16 six-term form (bf16 has fp32's exponent range) -- the forward stays finite and within tolerance of the CPU oracle
    on the same parameters, where an f16 split of the unbounded activations would overflow."""
    import torch
    from crowdmod_ddpm_4d_amd.unet import UNet
    from oracle import unet_torch as ot
    torch.set_num_threads(16)
    cfg = full_cfg(3)
    params = dict(spec.init_params(cfg, SEED_W))
    for k in ("encoder_blocks.0.normalize_2.weight", "bottleneck_blocks.1.normalize_1.weight", "final.0.weight"):
        params[k] = (np.asarray(params[k]) * 1000.0).astype(np.float32)
    n = UNet(input_channels=3, output_channels=3, num_res_blocks=1, base_channels=32, base_channels_multiples=(1, 2, 4),
             apply_attention=(False, False, True), dropout_rate=0.1, time_multiple=4, condition="Past", max_batch=B)
    n.load_state_dict(params)
    past, fut = synth_inputs(B, 3, 12, 36, 5, 3, "h2/fallback")
    t = np.array([5, 900])
    y = n(fut, t, past)
    plan = spec.make_plan(cfg)
    with torch.no_grad():
        ref = ot.unet_forward(ot.to_torch(params), plan, torch.from_numpy(fut), torch.from_numpy(t).long(), torch.from_numpy(past), None).numpy()
    scale = float(np.abs(ref).max())
    err = float(np.abs(y - ref).max())
    assert np.isfinite(y).all() and scale > 50.0            # the activations really are far outside the f16 range of an unscaled split
    assert err <= 2e-5 * scale, (err, scale)
