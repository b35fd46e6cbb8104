"""GPU parity tests AT the benchmarked configuration and on the other two reference grids
(round-1 verdict: the B = 64 ATC C = 4 workload of bench.py -- and with it the XCD-remapped tile
order, which only engages when gridDim.x % 8 == 0 -- was never compared with anything).

* ATC 12x36, C = 4, B = 64 and B = 8 forwards: samples 0, 1 against the reference's own outputs
  (fwd.npz), every sample bit-identical to the same sample run at B = 2 (the design claim behind the
  batch shard);
* a 20-step B = 64 reverse loop (the bench's exact call) bit-identical per chain to B = 2 loops;
* XCD tile remap on == off, bit for bit;
* the reference's own loops on the HERMES-CR-120 (28x24) and 2x ATC (24x72) grids (loop_grids.npz) and
  the full-width training step on CR-120 (train_full_cr120.npz).
Reference lines: models/backbones/unet.py:124-167, models/diffusion/ddpm.py:206-282,111-121.
"""
import ctypes as C

import numpy as np
import pytest

from crowdmod_ddpm_4d_amd import native, prng, spec
from helpers import FULL_GRIDS, SEED_W, full_cfg, load, loop_noise, synth_inputs

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: <= 1e-4 max-abs vs reference (fp32)
B_BENCH = 64


def _unet(C_, max_batch):
    from crowdmod_ddpm_4d_amd.unet import UNet
    cfg = full_cfg(C_)
    net = UNet(cfg.input_channels, cfg.output_channels, cfg.num_res_blocks, cfg.base_channels,
               cfg.base_channels_multiples, cfg.apply_attention, cfg.dropout_rate, cfg.time_multiple, "Past",
               max_batch=max_batch)
    net.load_state_dict(spec.init_params(cfg, SEED_W))
    return net


def _bench_inputs(C_=4):
    """64 samples on the ATC grid: 0, 1 are the fixture's inputs (fwd.npz atc_c4), the rest fresh draws."""
    H, W = FULL_GRIDS["atc"]
    g = load("fwd.npz")
    p2, f2 = synth_inputs(2, C_, H, W, 5, 3, f"full/atc/c{C_}")
    pr, fr = synth_inputs(B_BENCH - 2, C_, H, W, 5, 3, f"benchcfg/c{C_}")
    past, fut = np.concatenate([p2, pr]), np.concatenate([f2, fr])
    t = np.concatenate([g[f"atc_c{C_}/t"], (np.arange(B_BENCH - 2, dtype=np.int64) * 37 + 5) % 1000])
    return past, fut, t, g[f"atc_c{C_}/out"]


def test_bench_config_forward_b64_and_b8_vs_reference_and_b2():
    past, fut, t, ref01 = _bench_inputs()
    net64 = _unet(4, B_BENCH)
    y64 = net64(fut, t, past)
    assert np.isfinite(y64).all()
    assert float(np.abs(y64[:2] - ref01).max()) <= TOL
    y8 = net64(fut[:8], t[:8], past[:8])
    assert np.array_equal(y8, y64[:8])
    # the same samples two at a time through a handle built for max_batch 2 (54-workgroup launches: the
    # un-remapped tile order) and through the B = 64 handle at batch 2
    net2 = _unet(4, 2)
    for i in range(0, B_BENCH, 2):
        y2 = net2(fut[i:i + 2], t[i:i + 2], past[i:i + 2])
        assert np.array_equal(y2, y64[i:i + 2]), i
    y2b = net64(fut[10:12], t[10:12], past[10:12])
    assert np.array_equal(y2b, y64[10:12])
    assert float(np.abs(net2(fut[:2], t[:2], past[:2]) - ref01).max()) <= TOL


def test_xcd_tile_remap_on_equals_off():
    """dbg bit 4096 switches the XCD-aware block-id -> tile remap off (cm_conv.hip / cm_conv_small.hip): the
    result may not depend on it.  B = 64: every full-resolution launch has gridDim.x % 8 == 0."""
    past, fut, t, ref01 = _bench_inputs()
    net = _unet(4, B_BENCH)
    L = native.lib()
    try:
        native.check(L.cm_debug_conv_flags(4096))
        off = net(fut, t, past)
        native.check(L.cm_debug_conv_flags(0))
        on = net(fut, t, past)
    finally:
        native.check(L.cm_debug_conv_flags(-1))
    assert np.array_equal(on, off)
    assert float(np.abs(on[:2] - ref01).max()) <= TOL


def _atc_model(C_, T=1000, batch=B_BENCH):
    from crowdmod_ddpm_4d_amd.config import AttrDict
    from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model
    H, W = FULL_GRIDS["atc"]
    cfg = AttrDict({
        "MACROPROPS": {"ROWS": H, "COLS": W}, "DATASET": {"PAST_LEN": 5, "FUTURE_LEN": 3, "BATCH_SIZE": batch},
        "MODEL": {"NSAMPLES": batch, "NSAMPLES4PLOTS": 2, "DDPM": {
            "SAMPLER": "DDPM", "TIMESTEPS": T, "SCALE": 0.5, "SIGMA": 0.001, "DDIM_DIVIDER": 2,
            "GUIDANCE": "None", "LAMBDA_GUIDANCE": 0.0,
            "UNET": {"CONDITION": "Past", "NUM_RES_BLOCKS": 1, "BASE_CH": 32, "BASE_CH_MULT": [1, 2, 4],
                     "APPLY_ATTENTION": [False, False, True, False], "DROPOUT_RATE": 0.1, "TIME_EMB_MULT": 4}}}})
    m = DDPM_model(cfg, "DDPM-UNet", C_)
    m.denoiser.load_state_dict(spec.init_params(full_cfg(C_), SEED_W))
    return m


def test_bench_config_loop_b64_bit_identical_to_b2_chains():
    """bench.py's call: the first 20 steps of the T = 1000 DDPM loop at B = 64 with device-drawn noise; every
    chain must equal the same chain (global sample index) run in a B = 2 loop."""
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    past, _, _, _ = _bench_inputs()
    sampler = DDPM(timesteps=1000, scale=0.5)
    m64 = _atc_model(4)
    m64._sample_calls = 0
    x64, _ = m64._generate_ddpm(past, sampler, B_BENCH, sample_id_base=0, first_steps=20)
    assert np.isfinite(x64).all() and float(np.abs(x64).std()) > 0.1
    m2 = _atc_model(4, batch=2)
    for i in (0, 2, 30, 62):
        m2._sample_calls = 0
        x2, _ = m2._generate_ddpm(past[i:i + 2], sampler, 2, sample_id_base=i, first_steps=20)
        assert np.array_equal(x2, x64[i:i + 2]), i
    # and the health check stays quiet on a healthy loop, loud on a poisoned checkpoint
    o = m64._opts(native.SAMPLER_DDPM, first_steps=3)
    o.check_finite = 1
    x, _ = m64._run_loop(past, sampler, B_BENCH, o, False)
    assert np.isfinite(x).all()
    sd = m64.denoiser.state_dict()
    sd["final.2.bias"] = np.full_like(sd["final.2.bias"], np.nan)
    m64.denoiser.load_state_dict(sd)
    with pytest.raises(native.NativeError, match="non-finite"):
        m64._run_loop(past, sampler, B_BENCH, o, False)
    o.check_finite = 0
    x, _ = m64._run_loop(past, sampler, B_BENCH, o, False)       # unchecked: rc 0 with NaNs, as before
    assert np.isnan(x).any()


def _grid_model(gname, T, sampler="DDPM", divider=2, sigma=0.001):
    from crowdmod_ddpm_4d_amd.config import AttrDict
    from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model
    H, W = FULL_GRIDS[gname]
    cfg = AttrDict({
        "MACROPROPS": {"ROWS": H, "COLS": W}, "DATASET": {"PAST_LEN": 5, "FUTURE_LEN": 3, "BATCH_SIZE": 2},
        "MODEL": {"NSAMPLES": 2, "NSAMPLES4PLOTS": 2, "DDPM": {
            "SAMPLER": sampler, "TIMESTEPS": T, "SCALE": 0.5, "SIGMA": sigma, "DDIM_DIVIDER": divider,
            "GUIDANCE": "None", "LAMBDA_GUIDANCE": 0.0,
            "UNET": {"CONDITION": "Past", "NUM_RES_BLOCKS": 1, "BASE_CH": 32, "BASE_CH_MULT": [1, 2, 4],
                     "APPLY_ATTENTION": [False, False, True, False], "DROPOUT_RATE": 0.1, "TIME_EMB_MULT": 4}}}})
    m = DDPM_model(cfg, "DDPM-UNet", 3)
    m.denoiser.load_state_dict(spec.init_params(full_cfg(3), SEED_W))
    return m


def _grid_loop_inputs(tag, gname):
    C_, P_, F, B = 3, 5, 3, 2
    H, W = FULL_GRIDS[gname]
    per = C_ * H * W * F
    past = prng.normal(7, f"past/loop/{tag}", B * C_ * H * W * P_).reshape(B, C_, H, W, P_)
    x_T = prng.normal_per_sample(7, f"xT/{tag}", np.arange(B), per).reshape(B, C_, H, W, F)
    return past, x_T, per, (B, C_, H, W, F)


@pytest.mark.parametrize("gname", ["cr120", "atc2x"])
def test_loop_ddpm50_on_other_grids_vs_reference(gname):
    """The reference's own _generate_ddpm (ddpm.py:206-236), T = 50, on the HERMES-CR-120 28x24 grid
    (BASELINE configs[3]) and the doubled ATC grid 24x72 (configs[4])."""
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    g = load("loop_grids.npz")
    tag, T = f"{gname}_ddpm50", 50
    past, x_T, per, shape = _grid_loop_inputs(tag, gname)
    noise = np.stack([loop_noise(tag, 2, per, t).reshape(shape) for t in range(T - 1, 0, -1)])
    m = _grid_model(gname, T)
    x, _ = m._generate_ddpm(past, DDPM(timesteps=T, scale=0.5), 2, x_T=x_T, noise=noise)
    err = float(np.abs(x - g[tag + "/x0"]).max())
    assert err <= TOL, err


def test_loop_ddim_div50_on_cr120_vs_reference():
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    g = load("loop_grids.npz")
    tag = "cr120_ddim1000_div50"
    past, x_T, per, shape = _grid_loop_inputs(tag, "cr120")
    taus = np.arange(0, 999, 50)
    noise = np.stack([loop_noise(tag, 2, per, int(t)).reshape(shape) for t in reversed(taus)])
    m = _grid_model("cr120", 1000, sampler="DDIM", divider=50)
    x, _ = m._generate_ddim(past, taus, DDPM(timesteps=1000, scale=0.5), 2, x_T=x_T, noise=noise)
    ref = g[tag + "/x0"]
    err = float(np.abs(x - ref).max() / max(1.0, np.abs(ref).max()))   # |x|max ~ 47 on random-init weights
    assert err <= TOL, err


def test_training_step_full_width_cr120_vs_reference():
    """Full-width training step on the CR-120 grid (its own tile shapes in forward, data-gradient and
    weight-gradient kernels): loss, all 168 gradient norms, two gradient corners vs the reference's autograd."""
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    g = load("train_full_cr120.npz")
    C_, B = 3, 2
    H, W = FULL_GRIDS["cr120"]
    ucfg = full_cfg(C_)
    net = _unet(C_, B)
    past, fut = synth_inputs(B, C_, H, W, 5, 3, "trainfull/cr120")
    eps = prng.normal(7, "trainfull/eps/cr120", fut.size).reshape(fut.shape)
    masks = {}
    for blk in spec.make_plan(ucfg).res_blocks():
        u = prng.uniform_pm1(7, f"dropfull/{blk.prefix}", B * blk.cout).reshape(B, blk.cout)
        masks[blk.prefix] = ((u * 0.5 + 0.5) >= 0.1).astype(np.float32) / np.float32(0.9)
    net.ensure(H, W, 5, 3, B)
    net.train_init(lr=5e-5, betas=(0.5, 0.999), weight_decay=0.003)
    sampler = DDPM(timesteps=1000, scale=0.5)
    loss = net.train_step(sampler._handle, fut, past, g["t"], eps, drop_masks=masks, apply_update=False)
    assert abs(loss - float(g["loss"])) <= 1e-5 * max(1.0, float(g["loss"]))
    n = 0
    for key in g.files:
        if key.startswith("gnorm/"):
            name, ref = key[6:], float(g[key])
            got = float(np.sqrt((net.grad(name).astype(np.float64) ** 2).sum()))
            assert abs(got - ref) <= 1e-3 * ref + 2e-7, (name, got, ref)
            n += 1
        if key.startswith("gslice/"):
            ref = g[key]
            got = net.grad(key[7:])[:4, :4]
            assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-8, key
    assert n == 168


def test_release_keeps_training_state_across_handle_recreation():
    """ensure() with a larger batch re-creates the native handle; trained master weights and the Adam
    moments must survive (they used to be dropped silently)."""
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    from helpers import NARROW, narrow_cfg
    from crowdmod_ddpm_4d_amd.unet import UNet
    C_ = 3
    H, W, P, F = NARROW["H"], NARROW["W"], NARROW["P"], NARROW["F"]
    past, fut = synth_inputs(4, C_, H, W, P, F, "release")
    eps = prng.normal(7, "release/eps", fut.size).reshape(fut.shape)
    t = np.array([5, 300, 600, 900], dtype=np.int64)
    sampler = DDPM(timesteps=1000, scale=0.5)

    def fresh(mb):
        net = UNet(C_, C_, 1, 8, (1, 2, 4), (False, False, True, False), 0.1, 4, "Past", max_batch=mb)
        net.load_state_dict(spec.init_params(narrow_cfg(C_), SEED_W))
        net.ensure(H, W, P, F, 2)
        net.train_init(lr=1e-3, betas=(0.5, 0.999), weight_decay=0.003)
        return net
    a, b = fresh(2), fresh(4)
    for net in (a, b):
        net.train_step(sampler._handle, fut[:2], past[:2], t[:2], eps[:2], seed=3, apply_update=True)
    # a: the batch grows past max_batch -> the handle is re-created between the steps; b: same handle
    la = a.train_step(sampler._handle, fut, past, t, eps, seed=4, apply_update=True) if a.ensure(H, W, P, F, 4) else None
    lb = b.train_step(sampler._handle, fut, past, t, eps, seed=4, apply_update=True)
    assert abs(la - lb) <= 1e-6 * max(1.0, abs(lb))
    a.sync_trained(); b.sync_trained()
    for k in ("first.weight", "decoder_blocks.7.conv_2.weight"):
        assert np.abs(a.state_dict()[k] - b.state_dict()[k]).max() <= 1e-7, k
    step = C.c_int32()
    native.check(native.lib().cm_train_opt_step(a._handle, C.byref(step), 0))
    assert step.value == 2


def test_issue_flops_split_is_consistent_with_the_executed_flops():
    """cm_model_issue_flops (bench.py's roofline numerator) against cm_model_exec_flops per class.  Default plan, inference-only
    handle: the h2 layers (Winograd, quarter-resolution, last conv: f16 two-way splits) issue THREE 16-bit products per
    fp32-equivalent product, the upsample layers (raw input: bf16 three-way splits) six; a handle that trains keeps the six-term
    form everywhere (fp32-instruction FLOPs + 16-bit FLOPs / 6 == executed); the relaxed plan issues 3x, the f16 plan 1x."""
    past, fut, t, _ = _bench_inputs()
    net = _unet(4, 8)
    net(fut[:8], t[:8], past[:8])
    L, h = native.lib(), net._handle
    ex, f32, b16 = (C.c_double * 8)(), (C.c_double * 8)(), (C.c_double * 8)()
    native.check(L.cm_model_exec_flops(h, 8, ex))
    native.check(L.cm_model_issue_flops(h, 8, f32, b16))
    for i in range(8):
        assert f32[i] + b16[i] / 6.0 <= ex[i] * (1 + 1e-9) and ex[i] <= (f32[i] + b16[i] / 3.0) * (1 + 1e-9)
    assert b16[0] / 3.0 > 0.8 * ex[0] and b16[0] / 6.0 < 0.6 * ex[0]     # mostly h2 layers; the upsample convs at 6x
    assert f32[0] > 0                          # stride-2 / first convs stay on fp32 instructions
    i32, i16 = net.conv3_issue_flops(8)
    assert i32 == f32[0] and i16 == b16[0]
    b16_h2 = b16[0]
    # ---- a handle that trains: an optimizer step leaves the h2 fragments behind (the device repack maintains the bf16 ones), so
    # the plan counts -- and would run -- the six-term form until the next inference entry point re-derives them (refresh_h2)
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    net.train_init(lr=1e-3, betas=(0.5, 0.999), weight_decay=0.003)
    h = net._handle
    native.check(L.cm_model_exec_flops(h, 8, ex))
    native.check(L.cm_model_issue_flops(h, 8, f32, b16))
    for i in range(8):
        assert abs(f32[i] + b16[i] / 6.0 - ex[i]) <= 1e-9 * max(1.0, ex[i])
    assert b16[0] / 6.0 > 0.8 * ex[0] and b16[0] > 1.5 * b16_h2
    sampler = DDPM(timesteps=1000, scale=0.5)
    eps = prng.normal(5, "issue/eps", fut[:8].size).reshape(fut[:8].shape)
    net.train_step(sampler._handle, fut[:8], past[:8], t[:8], eps, apply_update=True)      # the weights move (lr 1e-3)
    net.sync_trained()                                                                   # master weights -> state_dict, time-embedding table (the sampling path's contract)
    y_tr = net(fut[:8], t[:8], past[:8])                                                 # inference: h2 fragments re-derived from the master weights
    native.check(L.cm_model_issue_flops(net._handle, 8, f32, b16))
    assert b16[0] == b16_h2                                                              # ... and the plan is back on h2
    fresh = _unet(4, 8)
    fresh.load_state_dict(net.state_dict())
    y_fr = fresh(fut[:8], t[:8], past[:8])
    y_0 = _unet(4, 8)(fut[:8], t[:8], past[:8])
    assert float(np.abs(y_tr - y_0).max()) > 1e-4                                        # the step really changed the network
    assert float(np.abs(y_tr - y_fr).max()) <= 1e-6, float(np.abs(y_tr - y_fr).max())   # same weights, same fragments: same output
    net = _unet(4, 8)
    net.set_precision("f32r")
    net(fut[:8], t[:8], past[:8])
    h = net._handle
    native.check(L.cm_model_exec_flops(h, 8, ex))
    native.check(L.cm_model_issue_flops(h, 8, f32, b16))
    for i in range(8):
        assert abs(f32[i] + b16[i] / 3.0 - ex[i]) <= 1e-9 * max(1.0, ex[i])
    net.set_precision("f16")
    net(fut[:8], t[:8], past[:8])
    h = net._handle                            # (a new native handle: the precision is fixed at plan-build time)
    native.check(L.cm_model_exec_flops(h, 8, ex))
    native.check(L.cm_model_issue_flops(h, 8, f32, b16))
    for i in range(8):
        assert abs(f32[i] + b16[i] - ex[i]) <= 1e-9 * max(1.0, ex[i])
    assert b16[0] > 0.5 * ex[0]
