"""Reduced-precision plan (BASELINE configs[4]: doubled ATC grid, f16 matrix-core operands): f16 operands with
fp32 accumulation in the Winograd 3x3x3 layers, everything else fp32 (cm_model_set_precision).  The reference's
analogue is torch.amp.autocast("cuda") around the denoiser (models/diffusion/ddpm.py:116-120); the fixtures are
the reference's fp32 outputs, so the tolerances below are the f16 path's distance from exact fp32:
  * one forward (|eps_hat| ~ 1):   <= 2e-2 max-abs, <= 2e-3 RMS   (observed: see the assert messages)
  * 50-step DDPM loop (|x| ~ 5.7): <= 5e-2 max-abs
fp32 stays the default and is untouched by this switch (bit-identical to a handle that never saw it)."""
import numpy as np
import pytest

from crowdmod_ddpm_4d_amd import native, prng, spec
from helpers import FULL_GRIDS, SEED_W, full_cfg, load, loop_noise, synth_inputs

pytestmark = pytest.mark.gpu

F16_FWD_MAXABS, F16_FWD_RMS, F16_LOOP_MAXABS = 2e-2, 2e-3, 5e-2


def _unet(C_, max_batch=2, precision="f32"):
    from crowdmod_ddpm_4d_amd.unet import UNet
    cfg = full_cfg(C_)
    net = UNet(cfg.input_channels, cfg.output_channels, cfg.num_res_blocks, cfg.base_channels,
               cfg.base_channels_multiples, cfg.apply_attention, cfg.dropout_rate, cfg.time_multiple, "Past", max_batch=max_batch)
    net.load_state_dict(spec.init_params(cfg, SEED_W))
    return net.set_precision(precision)


@pytest.mark.parametrize("key", ["atc2x_c3", "atc_c4", "cr120_c3"])
def test_f16_forward_within_stated_tolerance_of_fp32_reference(key):
    g = load("fwd.npz")
    gname, c = key.split("_c")
    C_ = int(c)
    H, W = FULL_GRIDS[gname]
    past, fut = synth_inputs(2, C_, H, W, 5, 3, f"full/{gname}/c{C_}")
    ref = g[f"{key}/out"]
    y32 = _unet(C_)(fut, g[f"{key}/t"], past)
    y16 = _unet(C_, precision="f16")(fut, g[f"{key}/t"], past)
    assert float(np.abs(y32 - ref).max()) <= 1e-4
    err = np.abs(y16 - ref)
    assert np.isfinite(y16).all()
    assert float(err.max()) <= F16_FWD_MAXABS and float(np.sqrt((err ** 2).mean())) <= F16_FWD_RMS, \
        (key, float(err.max()), float(np.sqrt((err ** 2).mean())))
    assert float(err.max()) > 1e-6          # the switch really changes the arithmetic


def test_f16_loop_on_doubled_grid_and_precision_rules():
    from crowdmod_ddpm_4d_amd.config import AttrDict
    from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    g = load("loop_grids.npz")
    tag, T, gname = "atc2x_ddpm50", 50, "atc2x"
    H, W = FULL_GRIDS[gname]
    C_, B = 3, 2
    per = C_ * H * W * 3
    cfg = AttrDict({
        "MACROPROPS": {"ROWS": H, "COLS": W}, "DATASET": {"PAST_LEN": 5, "FUTURE_LEN": 3, "BATCH_SIZE": B},
        "MODEL": {"NSAMPLES": B, "NSAMPLES4PLOTS": 2, "DDPM": {
            "SAMPLER": "DDPM", "TIMESTEPS": T, "SCALE": 0.5, "SIGMA": 0.001, "DDIM_DIVIDER": 2, "GUIDANCE": "None", "LAMBDA_GUIDANCE": 0.0,
            "UNET": {"CONDITION": "Past", "NUM_RES_BLOCKS": 1, "BASE_CH": 32, "BASE_CH_MULT": [1, 2, 4],
                     "APPLY_ATTENTION": [False, False, True, False], "DROPOUT_RATE": 0.1, "TIME_EMB_MULT": 4}}}})
    m = DDPM_model(cfg, "DDPM-UNet", C_)
    m.denoiser.load_state_dict(spec.init_params(full_cfg(C_), SEED_W))
    m.denoiser.set_precision("f16")
    past = prng.normal(7, f"past/loop/{tag}", B * C_ * H * W * 5).reshape(B, C_, H, W, 5)
    x_T = prng.normal_per_sample(7, f"xT/{tag}", np.arange(B), per).reshape(B, C_, H, W, 3)
    noise = np.stack([loop_noise(tag, B, per, t).reshape(x_T.shape) for t in range(T - 1, 0, -1)])
    x, _ = m._generate_ddpm(past, DDPM(timesteps=T, scale=0.5), B, x_T=x_T, noise=noise)
    err = float(np.abs(x - g[tag + "/x0"]).max())
    assert np.isfinite(x).all() and err <= F16_LOOP_MAXABS, err
    # training needs an fp32 handle (the f16 operand copies are not re-packed after an optimizer step)
    with pytest.raises(native.NativeError, match="fp32"):
        m.denoiser.train_init()
    # precision must be chosen before finalize; unknown values are rejected
    L = native.lib()
    with pytest.raises(native.NativeError, match="before"):
        native.check(L.cm_model_set_precision(m.denoiser._handle, 0))
    with pytest.raises(ValueError):
        m.denoiser.set_precision("bf16")


def test_f16_attention_core_on_matrix_cores_agrees_with_the_fp32_core():
    """216 tokens (24x72 grid) run the generic attention chain; under the reduced-precision plan its core -- QK^T, softmax,
    PV -- takes f16 matrix-core operands with fp32 accumulation and an fp32 softmax (attn_core_f16_kernel; the reference's
    autocast covers nn.MultiheadAttention, ddpm.py:116-120 / layers.py:16).  Against the fp32 plan's exact core on the same
    inputs: the first attention block's core output within 2e-2 of its own scale (both plans feed it slightly different
    q, k, v: the layers before it already differ by the f16 convolutions), finite, and not identical."""
    H, W = FULL_GRIDS["atc2x"]
    C_ = 3
    g = load("fwd.npz")
    past, fut = synth_inputs(2, C_, H, W, 5, 3, f"full/atc2x/c{C_}")
    n32, n16 = _unet(C_), _unet(C_, precision="f16")
    n32(fut, g["atc2x_c3/t"], past)
    n16(fut, g["atc2x_c3/t"], past)
    name = "encoder_blocks.4.attention.core"
    a32, a16 = n32.debug_activation(name)[:2], n16.debug_activation(name)[:2]
    assert a32.shape == a16.shape and np.isfinite(a16).all()
    scale = float(np.abs(a32).max())
    err = float(np.abs(a16 - a32).max())
    assert 0.0 < err <= 2e-2 * max(1.0, scale), (err, scale)
