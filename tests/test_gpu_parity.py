"""GPU parity tests: the HIP path (through the C ABI) vs the golden vectors captured
from the reference and vs the oracle.  Run on the MI355X box with `-m gpu`."""
import ctypes as C
import os

import numpy as np
import pytest

from crowdmod_ddpm_4d_amd import prng, spec
from helpers import FULL_GRIDS, NARROW, SEED_W, full_cfg, load, loop_noise, narrow_cfg, synth_inputs

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: <= 1e-4 max-abs vs reference (fp32)


def _unet(cfg, params, max_batch=2):
    from crowdmod_ddpm_4d_amd.unet import UNet
    net = UNet(cfg.input_channels, cfg.output_channels, cfg.num_res_blocks, cfg.base_channels,
               cfg.base_channels_multiples, cfg.apply_attention, cfg.dropout_rate, cfg.time_multiple, "Past",
               max_batch=max_batch)
    net.load_state_dict(params)
    return net


@pytest.mark.parametrize("C_", [3, 4])
def test_narrow_forward_every_block(C_):
    g = load(f"ops_c{C_}.npz")
    cfg = narrow_cfg(C_)
    P = spec.init_params(cfg, SEED_W)
    geo = NARROW
    past, fut = synth_inputs(geo["B"], C_, geo["H"], geo["W"], geo["P"], geo["F"], f"narrow{C_}")
    net = _unet(cfg, P)
    y = net(fut, g["t"], past)
    plan = spec.make_plan(cfg)
    names = ["first"] + [b.prefix for b in plan.encoder + plan.bottleneck + plan.decoder]
    worst = {}
    for name in names:
        act = net.debug_activation(name)[: geo["B"]]
        ref = g[name + "/out"]
        assert act.shape == ref.shape, (name, act.shape, ref.shape)
        worst[name] = float(np.abs(act - ref).max())
    bad = {k: v for k, v in worst.items() if v > TOL}
    assert not bad, f"blocks beyond tolerance: {bad} (all: {worst})"
    assert np.abs(y - g["out"]).max() <= TOL


@pytest.mark.parametrize("key", ["atc_c3", "atc_c4", "cr120_c3", "atc2x_c3"])
def test_full_forward_vs_reference(key):
    g = load("fwd.npz")
    gname, c = key.split("_c")
    C_ = int(c)
    H, W = FULL_GRIDS[gname]
    cfg = full_cfg(C_)
    P = spec.init_params(cfg, SEED_W)
    past, fut = synth_inputs(2, C_, H, W, 5, 3, f"full/{gname}/c{C_}")
    net = _unet(cfg, P)
    y = net(fut, g[f"{key}/t"], past)
    err = float(np.abs(y - g[f"{key}/out"]).max())
    assert err <= TOL, err


def test_forward_batch_independence_and_odd_batch():
    """Chains are independent: sample i of a batch of 5 equals sample i run alone
    (the property the multi-GPU batch shard relies on)."""
    cfg = narrow_cfg(3)
    P = spec.init_params(cfg, SEED_W)
    past, fut = synth_inputs(5, 3, 4, 8, 5, 3, "odd")
    t = np.array([0, 1, 500, 998, 999], dtype=np.int64)
    net = _unet(cfg, P, max_batch=5)
    full = net(fut, t, past)
    for i in (0, 4):
        one = net(fut[i:i + 1], t[i:i + 1], past[i:i + 1])
        assert np.array_equal(one[0], full[i]), i


def _loop_inputs(tag):
    C_, H, W, P_, F, B = 3, 12, 36, 5, 3, 2
    per = C_ * H * W * F
    past = prng.normal(7, f"past/loop/{tag}", B * C_ * H * W * P_).reshape(B, C_, H, W, P_)
    x_T = prng.normal_per_sample(7, f"xT/{tag}", np.arange(B), per).reshape(B, C_, H, W, F)
    return past, x_T, per, (B, C_, H, W, F)


def _model(T, guidance="None", lam=0.0, sigma=0.001, divider=2, sampler="DDPM"):
    from crowdmod_ddpm_4d_amd.config import AttrDict
    from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model
    cfg = AttrDict({
        "MACROPROPS": {"ROWS": 12, "COLS": 36}, "DATASET": {"PAST_LEN": 5, "FUTURE_LEN": 3, "BATCH_SIZE": 2},
        "MODEL": {"NSAMPLES": 2, "NSAMPLES4PLOTS": 2, "DDPM": {
            "SAMPLER": sampler, "TIMESTEPS": T, "SCALE": 0.5, "SIGMA": sigma, "DDIM_DIVIDER": divider,
            "GUIDANCE": guidance, "LAMBDA_GUIDANCE": lam,
            "UNET": {"CONDITION": "Past", "NUM_RES_BLOCKS": 1, "BASE_CH": 32, "BASE_CH_MULT": [1, 2, 4],
                     "APPLY_ATTENTION": [False, False, True, False], "DROPOUT_RATE": 0.1, "TIME_EMB_MULT": 4}}}})
    m = DDPM_model(cfg, "DDPM-UNet", 3)
    m.denoiser.load_state_dict(spec.init_params(full_cfg(3), SEED_W))
    return m


@pytest.mark.parametrize("tag,T,guid", [("ddpm50", 50, "None"), ("ddpm20_sparsity", 20, "Sparsity")])
def test_loop_ddpm_vs_reference(tag, T, guid):
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    g = load("loop.npz")
    past, x_T, per, shape = _loop_inputs(tag)
    noise = np.stack([loop_noise(tag, 2, per, t).reshape(shape) for t in range(T - 1, 0, -1)])
    m = _model(T, guid, 0.004)
    x, lst = m._generate_ddpm(past, DDPM(timesteps=T, scale=0.5), 2, x_T=x_T, noise=noise)
    err = float(np.abs(x - g[tag + "/x0"]).max())
    assert err <= TOL, err
    assert lst[0] is x_T and lst[-1] is x


@pytest.mark.parametrize("guid", ["None", "Sparsity"])
def test_loop_ddim_vs_reference(guid):
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    g = load("loop.npz")
    tag = "ddim1000_div100" + ("_sparsity" if guid == "Sparsity" else "")
    past, x_T, per, shape = _loop_inputs(tag)
    taus = np.arange(0, 999, 100)
    noise = np.stack([loop_noise(tag, 2, per, int(t)).reshape(shape) for t in reversed(taus)])
    m = _model(1000, guid, 0.004, sigma=0.001, divider=100, sampler="DDIM")
    x, _ = m._generate_ddim(past, taus, DDPM(timesteps=1000, scale=0.5), 2, x_T=x_T, noise=noise)
    ref = g[tag + "/x0"]
    err = float(np.abs(x - ref).max() / max(1.0, np.abs(ref).max()))
    assert err <= TOL, err


def test_loop_ddpm1000_history_vs_reference():
    """Full-length loop (BASELINE config 2 length, B=2) with history; bar relative to
    |x|max because the reference itself moves by 5e-5 with the thread count here."""
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    g = load("loop.npz")
    tag, T = "ddpm1000", 1000
    past, x_T, per, shape = _loop_inputs(tag)
    noise = np.stack([loop_noise(tag, 2, per, t).reshape(shape) for t in range(T - 1, 0, -1)])
    m = _model(T)
    x, hist = m._generate_ddpm(past, DDPM(timesteps=T, scale=0.5), 2, history=True, x_T=x_T, noise=noise)
    assert len(hist) == T + 1 and np.array_equal(hist[0], x_T) and np.array_equal(hist[-1], x)
    for t in (999, 900, 500, 0):
        ref = g[f"{tag}/x_after_t{t}"]
        got = hist[1 + (T - 1 - t)]
        rel = float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))
        assert rel <= 2e-5, (t, rel)


def test_schedule_q_sample_and_step_vs_golden():
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    from oracle import unet_numpy as on
    g = load("schedule.npz")
    s = DDPM(timesteps=1000, scale=0.5)
    for k in ("beta", "alpha", "alpha_bar"):
        assert np.array_equal(getattr(s, k), g[f"T1000_s0.5/{k}"]), k
    for k in ("sqrt_alpha_bar", "one_by_sqrt_alpha", "sqrt_one_minus_alpha_bar"):
        np.testing.assert_allclose(getattr(s, k), g[f"T1000_s0.5/{k}"], rtol=1.2e-7)
    x0 = prng.normal(7, "qs/x0", 4 * 3 * 4 * 8 * 3).reshape(4, 3, 4, 8, 3)
    eps = prng.normal(7, "qs/eps", 4 * 3 * 4 * 8 * 3).reshape(4, 3, 4, 8, 3)
    xt, e = s(x0, g["qsample/t"], noise=eps)
    np.testing.assert_allclose(xt, g["qsample/xt"], rtol=0, atol=1e-6)
    sched = on.schedule(1000, 0.5)
    for t in (999, 500, 1, 0):
        want, sig, a = on.ddpm_step(sched, eps, x0, t, x0[::-1].copy() if t > 0 else np.zeros_like(x0))
        got, sig2, a2 = s.step(eps, x0, t, noise=x0[::-1].copy())
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
        assert abs(sig - sig2) < 1e-7 and abs(a - a2) < 1e-7


def test_device_rng_is_shard_independent_and_normal():
    """x_T / z_t drawn on the device depend only on (seed, global sample id, step)."""
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    m = _model(4)
    past, _, per, shape = _loop_inputs("rng")
    past4 = np.concatenate([past, past[::-1]])
    s = DDPM(timesteps=4, scale=0.5)
    m.seed = 123
    m._sample_calls = 0
    full, hist = m._generate_ddpm(past4, s, 4, history=True)
    m._sample_calls = 0
    lo, _ = m._generate_ddpm(past4[:2], s, 2, sample_id_base=0)
    m._sample_calls = 0
    hi, _ = m._generate_ddpm(past4[2:], s, 2, sample_id_base=2)
    assert np.array_equal(full[:2], lo) and np.array_equal(full[2:], hi)
    xT = hist[0]
    assert abs(float(xT.mean())) < 0.05 and abs(float(xT.std()) - 1.0) < 0.05


def test_training_forward_and_loss_vs_reference():
    """Forward half of DDPM_model._train_step (ddpm.py:111-121) on the narrow model: q-sample,
    UNet in train mode with the Dropout3d masks injected, MSE -- against the loss and the
    prediction the reference produced for the same t / eps / masks."""
    from crowdmod_ddpm_4d_amd.config import AttrDict
    from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    g = load("train.npz")
    C_, B = 3, 4
    cfg = AttrDict({
        "MACROPROPS": {"ROWS": NARROW["H"], "COLS": NARROW["W"]},
        "DATASET": {"PAST_LEN": NARROW["P"], "FUTURE_LEN": NARROW["F"], "BATCH_SIZE": B},
        "MODEL": {"DDPM": {"TIMESTEPS": 1000, "SCALE": 0.5, "UNET": {
            "CONDITION": "Past", "NUM_RES_BLOCKS": 1, "BASE_CH": 8, "BASE_CH_MULT": [1, 2, 4],
            "APPLY_ATTENTION": [False, False, True, False], "DROPOUT_RATE": 0.1, "TIME_EMB_MULT": 4}}}})
    m = DDPM_model(cfg, "DDPM-UNet", C_)
    ucfg = narrow_cfg(C_)
    m.denoiser.load_state_dict(spec.init_params(ucfg, SEED_W))
    past, fut = synth_inputs(B, C_, NARROW["H"], NARROW["W"], NARROW["P"], NARROW["F"], "train")
    eps = prng.normal(7, "train/eps", fut.size).reshape(fut.shape)
    masks = {}
    for blk in spec.make_plan(ucfg).res_blocks():
        u = prng.uniform_pm1(7, f"drop/{blk.prefix}", B * blk.cout).reshape(B, blk.cout)
        masks[blk.prefix] = ((u * 0.5 + 0.5) >= 0.1).astype(np.float32) / np.float32(0.9)
    loss, pred = m._train_step(fut, past, DDPM(timesteps=1000, scale=0.5), t=g["t"], noise=eps, drop_masks=masks)
    assert np.abs(pred - g["pred"]).max() <= TOL
    assert abs(loss - float(g["loss"])) <= 1e-5 * max(1.0, float(g["loss"]))
    # masks drawn on the device: keep-probability and scaling are right, values are {0, 1/(1-p)}
    m.denoiser.train()
    p2 = m.denoiser.forward_train(fut, g["t"], past, seed=5)
    p3 = m.denoiser.forward_train(fut, g["t"], past, seed=5)
    m.denoiser.eval()
    assert np.array_equal(p2, p3) and not np.array_equal(p2, pred)


def _narrow_train_setup():
    from crowdmod_ddpm_4d_amd.config import AttrDict
    from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    C_, B = 3, 4
    cfg = AttrDict({
        "MACROPROPS": {"ROWS": NARROW["H"], "COLS": NARROW["W"]},
        "DATASET": {"PAST_LEN": NARROW["P"], "FUTURE_LEN": NARROW["F"], "BATCH_SIZE": B},
        "MODEL": {"DDPM": {"TIMESTEPS": 1000, "SCALE": 0.5, "UNET": {
            "CONDITION": "Past", "NUM_RES_BLOCKS": 1, "BASE_CH": 8, "BASE_CH_MULT": [1, 2, 4],
            "APPLY_ATTENTION": [False, False, True, False], "DROPOUT_RATE": 0.1, "TIME_EMB_MULT": 4}}}})
    m = DDPM_model(cfg, "DDPM-UNet", C_)
    ucfg = narrow_cfg(C_)
    m.denoiser.load_state_dict(spec.init_params(ucfg, SEED_W))
    past, fut = synth_inputs(B, C_, NARROW["H"], NARROW["W"], NARROW["P"], NARROW["F"], "train")
    eps = prng.normal(7, "train/eps", fut.size).reshape(fut.shape)
    masks = {}
    for blk in spec.make_plan(ucfg).res_blocks():
        u = prng.uniform_pm1(7, f"drop/{blk.prefix}", B * blk.cout).reshape(B, blk.cout)
        masks[blk.prefix] = ((u * 0.5 + 0.5) >= 0.1).astype(np.float32) / np.float32(0.9)
    return m, DDPM(timesteps=1000, scale=0.5), past, fut, eps, masks


@pytest.mark.gpu
def test_training_backward_gradients_vs_reference():
    """loss.backward() of ddpm.py:142-143 on the narrow model: every parameter's gradient norm and six
    full gradient tensors (first / final conv, attention in_proj, a 3x3x3 conv, a GroupNorm weight over a
    channel concat, the time MLP) against the reference's autograd for the same t / eps / masks."""
    g = load("train.npz")
    m, sampler, past, fut, eps, masks = _narrow_train_setup()
    net = m.denoiser
    net.ensure(NARROW["H"], NARROW["W"], NARROW["P"], NARROW["F"], 4)
    net.train_init(lr=5e-5, betas=(0.5, 0.999), weight_decay=0.003)
    loss = net.train_step(sampler._handle, fut, past, g["t"], eps, drop_masks=masks, apply_update=False)
    assert abs(loss - float(g["loss"])) <= 1e-5 * max(1.0, float(g["loss"]))
    worst = []
    for key in g.files:
        if key.startswith("grad/"):
            name = key[5:]
            ref = g[key]
            got = net.grad(name)
            err = float(np.abs(got - ref).max())
            # fp32 accumulation over up to ~1e4 terms in a different order than autograd's
            assert err <= 2e-4 * float(np.abs(ref).max()) + 1e-8, (name, err, float(np.abs(ref).max()))
        if key.startswith("gnorm/"):
            name = key[6:]
            ref = float(g[key])
            got = float(np.sqrt((net.grad(name).astype(np.float64) ** 2).sum()))
            worst.append((abs(got - ref) / (ref + 1e-6), name, got, ref))
            # tensors whose true gradient vanishes (bias / time projection in front of a per-channel
            # GroupNorm) hold rounding noise of order 1e-8 on both sides: absolute floor
            assert abs(got - ref) <= 1e-3 * ref + 2e-7, (name, got, ref)
    assert len(worst) == 168  # every state_dict tensor except the frozen sinusoid table
    # the flat gradient buffer the data-parallel all-reduce works on: zero-copy torch view (RCCL reduces
    # this tensor in place), laid out in state_dict order
    import torch
    from crowdmod_ddpm_4d_amd.distributed import _DevView
    ptr, n = net.flat_grads()
    view = torch.as_tensor(_DevView(ptr, n), device="cuda:0")
    assert view.data_ptr() == ptr and view.numel() == sum(v.size for v in net.state_dict().values())
    off = 0
    for name, v in net.state_dict().items():
        if name == "final.2.weight":
            assert np.array_equal(view[off:off + v.size].cpu().numpy().reshape(v.shape), net.grad(name))
        off += v.size


@pytest.mark.gpu
def test_training_two_adam_steps_vs_reference():
    """optimizer.step() (torch.optim.Adam with coupled weight decay, ddpm.py:53-56,144) twice on the same
    batch: updated weights of nine tensors after step 1 and step 2, and the loss of the second forward
    (which runs on the re-packed weights), against the reference."""
    g = load("train.npz")
    m, sampler, past, fut, eps, masks = _narrow_train_setup()
    net = m.denoiser
    net.ensure(NARROW["H"], NARROW["W"], NARROW["P"], NARROW["F"], 4)
    net.train_init(lr=5e-5, betas=(0.5, 0.999), weight_decay=0.003)
    lr = 5e-5
    for step in (1, 2):
        loss = net.train_step(sampler._handle, fut, past, g["t"], eps, drop_masks=masks, apply_update=True)
        ref_loss = float(g["loss"] if step == 1 else g["loss2"])
        assert abs(loss - ref_loss) <= 2e-5 * max(1.0, ref_loss), (step, loss, ref_loss)
        net.sync_trained()
        sd = net.state_dict()
        for key in g.files:
            if not key.startswith(f"post{step}/"):
                continue
            name = key.split("/", 1)[1]
            diff = np.abs(sd[name] - g[key])
            # an Adam step moves every element by ~lr * sign(g): elements whose gradient nearly cancels the
            # weight-decay term may land anywhere within one step; all others agree to rounding
            assert float(diff.max()) <= 2.05 * lr * step, (name, float(diff.max()))
            assert float((diff > 2e-7).mean()) <= 0.002, (name, float((diff > 2e-7).mean()))
    # eval-mode forward runs on the trained weights (time table rebuilt, weights re-packed)
    t = g["t"]
    out = net(fut, t, past)
    from oracle import unet_numpy as onp
    ref = onp.unet_forward(sd, spec.make_plan(narrow_cfg(3)), fut, t, past)
    assert np.abs(out - ref).max() <= TOL


@pytest.mark.gpu
def test_train_loop_checkpoint_and_resume(tmp_path):
    """DDPM_model.train (ddpm.py:156-202) on a small synthetic set: the loss falls, the best-loss
    checkpoint is written under the reference's file name in the reference's {"opt","model"} format,
    torch reads it back, and a fresh handle resumed from it (weights + Adam state) continues bit-exactly."""
    import torch
    from crowdmod_ddpm_4d_amd import checkpoint
    from crowdmod_ddpm_4d_amd.config import AttrDict
    from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    B, C_ = 8, 3
    H, W, P, F = NARROW["H"], NARROW["W"], NARROW["P"], NARROW["F"]

    def make():
        cfg = AttrDict({
            "MACROPROPS": {"ROWS": H, "COLS": W},
            "DATASET": {"NAME": "synthetic", "PAST_LEN": P, "FUTURE_LEN": F, "BATCH_SIZE": B},
            "DATA_FS": {"SAVE_DIR": str(tmp_path) + "/"},
            "MODEL": {"NAME": "{}_SYN_TE{}_PL{}_FL{}_CE{}_{}.pth", "DDPM": {"TIMESTEPS": 1000, "SCALE": 0.5,
                      "CHECKPOINTS_TO_KEEP": 1, "UNET": {
                          "CONDITION": "Past", "NUM_RES_BLOCKS": 1, "BASE_CH": 8, "BASE_CH_MULT": [1, 2, 4],
                          "APPLY_ATTENTION": [False, False, True, False], "DROPOUT_RATE": 0.1, "TIME_EMB_MULT": 4,
                          "TRAIN": {"EPOCHS": 4, "SOLVER": {"LR": 2e-3, "BETAS": [0.5, 0.999], "WEIGHT_DECAY": 0.003,
                                    "SCHEDULER": {"FACTOR": 0.5, "PATIENCE": 10, "MIN_LR": 1e-6}}}}}}})
        m = DDPM_model(cfg, "DDPM-UNet", C_)
        m.denoiser.load_state_dict(spec.init_params(narrow_cfg(C_), SEED_W))
        return m
    n = 6 * B
    past = prng.normal(3, "tl/past", n * C_ * H * W * P).reshape(n, C_, H, W, P)
    fut = prng.normal(3, "tl/fut", n * C_ * H * W * F).reshape(n, C_, H, W, F)
    loader = [(past[i:i + B], fut[i:i + B]) for i in range(0, n, B)]
    m = make()
    hist = m.train(loader)
    assert len(hist) == 4 and all(np.isfinite(hist)) and hist[-1] < 0.8 * hist[0], hist
    best = m.checkpoint_path("000")
    assert best.endswith("DDPM-UNet_SYN_TE4_PL%d_FL%d_CE000_NA.pth" % (P, F)) and os.path.isfile(best)
    ck = torch.load(best, map_location="cpu", weights_only=True)
    assert set(ck.keys()) == {"opt", "model"} and len(ck["model"]) == 169
    assert len(ck["opt"]["state"]) == 168 and ck["opt"]["param_groups"][0]["betas"] == (0.5, 0.999)
    sd = m.denoiser.state_dict()
    assert not np.array_equal(sd["final.2.weight"], spec.init_params(narrow_cfg(C_), SEED_W)["final.2.weight"])
    # resume: weights + Adam moments + step counter from the file, then one more identical step on both
    m.save_checkpoint("resume", path=str(tmp_path / "resume.pth"))
    ck = torch.load(str(tmp_path / "resume.pth"), map_location="cpu", weights_only=True)
    assert np.array_equal(ck["model"]["final.2.weight"].numpy(), sd["final.2.weight"])
    assert float(ck["opt"]["state"][5]["step"]) == 24.0
    m2 = make()
    m2.load_checkpoint(str(tmp_path / "resume.pth"))
    m2._ensure_training(past[:B], fut[:B])
    m2.denoiser.load_optimizer_state_dict(checkpoint.load(str(tmp_path / "resume.pth"))["opt"])
    sampler = DDPM(timesteps=1000, scale=0.5)
    t = np.arange(B, dtype=np.int64) * 100
    eps = prng.normal(3, "tl/eps", fut[:B].size).reshape(fut[:B].shape)
    out = []
    for mm in (m, m2):
        l = mm.denoiser.train_step(sampler._handle, fut[:B], past[:B], t, eps, seed=99, apply_update=True)
        mm.denoiser.sync_trained()
        out.append((l, mm.denoiser.state_dict()["encoder_blocks.0.conv_1.weight"].copy()))
    assert abs(out[0][0] - out[1][0]) <= 1e-6 * max(1.0, abs(out[0][0]))
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-7


def _fm_setup():
    from crowdmod_ddpm_4d_amd.config import AttrDict
    from crowdmod_ddpm_4d_amd.flow_matching import FM_model
    C_, B = 3, 2
    cfg = AttrDict({
        "MACROPROPS": {"ROWS": NARROW["H"], "COLS": NARROW["W"]},
        "DATASET": {"NAME": "synthetic", "PAST_LEN": NARROW["P"], "FUTURE_LEN": NARROW["F"], "BATCH_SIZE": B},
        "DATA_FS": {"SAVE_DIR": "/tmp/"},
        "MODEL": {"NAME": "{}_SYN_TE{}_PL{}_FL{}_CE{}_{}.pth", "NSAMPLES4PLOTS": B, "FM": {
            "TIME_MAX_POS": 1000, "CHECKPOINTS_TO_KEEP": 1, "W_TYPE": "Linear", "INTEGRATOR": "Euler",
            "INTEGRATOR_STEPS": {"EULER": 8, "HEUN": 4},
            "UNET": {"CONDITION": "Past", "NUM_RES_BLOCKS": 1, "BASE_CH": 8, "BASE_CH_MULT": [1, 2, 4],
                     "APPLY_ATTENTION": [False, False, True, False], "DROPOUT_RATE": 0.1, "TIME_EMB_MULT": 4,
                     "TRAIN": {"EPOCHS": 1, "SOLVER": {"LR": 1e-4, "BETAS": [0.5, 0.999], "WEIGHT_DECAY": 0.001,
                               "SCHEDULER": {"FACTOR": 0.5, "PATIENCE": 10, "MIN_LR": 1e-6}}}}}}})
    m = FM_model(cfg, "FM-UNet", C_)
    m.denoiser.load_state_dict(spec.init_params(narrow_cfg(C_), SEED_W))
    past, fut = synth_inputs(B, C_, NARROW["H"], NARROW["W"], NARROW["P"], NARROW["F"], "fm")
    x0 = prng.normal(7, "fm/x0", fut.size).reshape(fut.shape)
    return m, past, fut, x0


@pytest.mark.gpu
def test_flow_matching_euler_sampling_vs_reference():
    """FM_model.sampling_with_euler (flow_matching.py:203-224), 8 steps, x_0 injected: the whole integration
    runs as one device loop; against the reference's own routine on the narrow model."""
    g = load("fm.npz")
    m, past, fut, x0 = _fm_setup()
    x = m.sampling_with_euler(past, 2, x0=x0)
    assert np.abs(x - g["euler8"]).max() <= TOL
    assert m.integrators["Heun"] == m.integrators["Euler"]   # flow_matching.py:44-47 maps both to Euler
    # device-drawn x_0: deterministic per seed, different from the injected run
    m2, *_ = _fm_setup()
    a = m2.sampling_with_euler(past, 2)
    m3, *_ = _fm_setup()
    b = m3.sampling_with_euler(past, 2)
    assert np.array_equal(a, b) and not np.array_equal(a, x)


@pytest.mark.gpu
def test_flow_matching_train_step_vs_reference():
    """Body of _train_one_epoch_fm (flow_matching.py:128-146) for both probability paths with x_0, t and the
    Dropout3d masks injected: loss and gradient norms of the conv_2 / first / final tensors."""
    g = load("fm.npz")
    for wtype in ("Linear", "Conic"):
        m, past, fut, x0 = _fm_setup()
        m.w_type = wtype
        masks = {}
        for blk in spec.make_plan(narrow_cfg(3)).res_blocks():
            u = prng.uniform_pm1(7, f"drop/{blk.prefix}", 2 * blk.cout).reshape(2, blk.cout)
            masks[blk.prefix] = ((u * 0.5 + 0.5) >= 0.1).astype(np.float32) / np.float32(0.9)
        m._ensure_training(past, fut)
        t = g["t"]
        xt, u_target = m.w_type_fns[wtype](x0, fut, t.reshape(-1, 1, 1, 1, 1))
        t_idx = (t * np.float32(1000)).astype(np.int64)
        loss = m.denoiser.train_step_xt(xt.astype(np.float32), past, t_idx, u_target.astype(np.float32), drop_masks=masks,
                                        apply_update=False)
        ref = float(g[f"{wtype}/loss"])
        assert abs(loss - ref) <= 2e-5 * max(1.0, ref), (wtype, loss, ref)
        for key in g.files:
            if key.startswith(f"{wtype}/gnorm/"):
                name = key.split("/", 2)[2]
                got = float(np.sqrt((m.denoiser.grad(name).astype(np.float64) ** 2).sum()))
                assert abs(got - float(g[key])) <= 2e-3 * float(g[key]) + 2e-7, (wtype, name, got, float(g[key]))
    # one epoch through the mirror's own loop (device dropout, host-drawn x_0 / t): finite loss, weights move
    m, past, fut, _ = _fm_setup()
    before = m.denoiser.state_dict()["final.2.weight"].copy()
    l = m._train_one_epoch_fm([(past, fut)], 1)
    m.denoiser.sync_trained()
    assert np.isfinite(l) and not np.array_equal(before, m.denoiser.state_dict()["final.2.weight"])


@pytest.mark.gpu
def test_frame_metrics_vs_reference_tables():
    """Device reductions behind PSNR / masked PSNR / relative density / TV (metricsGenerator.py:70-92,120-186,
    293-339) against the tables the reference's MetricsGenerator produced for the same sequences, including
    a frame whose density mask is empty (NaN like the reference's mean over an empty selection)."""
    from crowdmod_ddpm_4d_amd.metrics import MetricsGenerator
    g = load("metrics.npz")
    N, C_, H, W, F, chunk = 8, 3, 12, 36, 3, int(g["chunk"])
    gt = prng.normal(7, "metrics/gt", N * C_ * H * W * F).reshape(N, C_, H, W, F)
    gt[:, 0] = np.maximum(gt[:, 0], 0.0)
    gt[0, 0, :, :, 2] = 0.0
    pred = (gt + 0.3 * prng.normal(7, "metrics/noise", gt.size).reshape(gt.shape)).astype(np.float32)
    for i in range(0, N, chunk):
        gt[i:i + chunk] = gt[i]
    mg = MetricsGenerator(pred, gt.astype(np.float32), 3)
    np.testing.assert_allclose(mg.ranges, g["ranges"], rtol=1e-6)
    eps = float(g["eps"])
    mg.compute_psnr_metric(chunk, eps, masked_flag=False)
    mg.compute_psnr_metric(chunk, eps, masked_flag=True)
    mg.compute_re_density_metric(chunk, eps)
    mg.compute_tv_metric()
    for k in ("PSNR", "MAX_PSNR", "PSNR_OVER_TIME", "MAX_PSNR_OVER_TIME", "MASK_PSNR", "MAX_MASK_PSNR", "MASK_PSNR_OVER_TIME",
              "MAX_MASK_PSNR_OVER_TIME", "RE_DENSITY", "MIN_RE_DENSITY"):
        np.testing.assert_allclose(mg.data_dict[k], g[k], rtol=2e-6, atol=1e-6, equal_nan=True, err_msg=k)
    # total variation: the reference sums |diff| in float32 (pairwise), the device in double
    np.testing.assert_allclose(mg.data_dict["TV_OVER_TIME"], g["TV_OVER_TIME"], rtol=0, atol=2e-3, err_msg="TV")
    assert np.isnan(g["MASK_PSNR_OVER_TIME"]).any()


@pytest.mark.gpu
def test_metric_dispatch_all_and_motion_feature_tables_through_the_generator(tmp_path):
    """compute_metrics (metricsGenerator.py:379-395) on the MetricsGenerator that owns the device reductions: 'ALL' fills the
    PSNR / SSIM / motion-feature / density / TV tables (never ENERGY: the reference lists it under 'ALLA'), the motion-feature
    tables equal the reference's (tests/golden/motion_feat.npz), both spellings of the Bhattacharyya metric select it, the CSV
    writer emits the reference's column headers, and an unknown metric is a ValueError."""
    from crowdmod_ddpm_4d_amd.metrics import MetricsGenerator, compute_metrics
    g = load("motion_feat.npz")
    mf = {"f": int(g["atc/fkg"][0]), "k": int(g["atc/fkg"][1]), "GAMMA": float(g["atc/fkg"][2])}
    mg = compute_metrics(MetricsGenerator(g["pred"], g["gt"], 3), "ALL", 2, 1e-8, motion_feature=mf)
    for k in ("PSNR", "MASK_PSNR", "SSIM", "MF_MSE", "MF_BHATT_DIST", "MF_BHATT_COEF", "RE_DENSITY", "TV_OVER_TIME"):
        assert mg.data_dict.get(k) is not None, k
    assert "ENERGY" not in mg.data_dict
    for k in ("MF_MSE", "MF_BHATT_DIST", "MF_BHATT_COEF"):
        np.testing.assert_allclose(mg.data_dict[k], g[f"atc/{k}"], rtol=1e-10, atol=1e-15, err_msg=k)
    for name in ("MF_BHATT", "MOTION_FEAT_BHATT"):
        one = compute_metrics(MetricsGenerator(g["pred"], g["gt"], 3), name, 2, 1e-8, motion_feature=mf)
        assert one.data_dict["MF_MSE"] is None and "PSNR" not in one.data_dict
        np.testing.assert_allclose(one.data_dict["MF_BHATT_COEF"], g["atc/MF_BHATT_COEF"], rtol=1e-10)
    index = mg.save_data_metrics(str(tmp_path), "t", 6)
    assert open(index["MF_BHATT_DIST"]).readline().strip() == "BHATT_DIST_Hist_2D_Based,BHATT_DIST_Hist_1D_Based"
    assert open(index["MF_MSE"]).readline().strip() == "MSE_Hist_2D_Based,MSE_Hist_1D_Based"
    with pytest.raises(ValueError):
        compute_metrics(mg, "MF_COSINE", 2, 1e-8)


@pytest.mark.gpu
def test_training_backward_full_width_model_vs_reference():
    """The same training step on the FULL-width model (base 32, ATC 12x36, B = 2): the 32-channel-chunk register-ring
    kernels, K-split quarter-resolution layers, parity-form upsample convs and 128-wide attention of the backward
    pass against the reference's autograd: loss, all 168 gradient norms, two gradient corners."""
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    from crowdmod_ddpm_4d_amd.unet import UNet
    g = load("train_full.npz")
    C_, B = 3, 2
    H, W = FULL_GRIDS["atc"]
    ucfg = full_cfg(C_)
    net = UNet(input_channels=C_, output_channels=C_, num_res_blocks=1, base_channels=32, base_channels_multiples=(1, 2, 4),
               apply_attention=(False, False, True), dropout_rate=0.1, time_multiple=4, condition="Past", max_batch=B)
    net.load_state_dict(spec.init_params(ucfg, SEED_W))
    past, fut = synth_inputs(B, C_, H, W, 5, 3, "trainfull")
    eps = prng.normal(7, "trainfull/eps", fut.size).reshape(fut.shape)
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


@pytest.mark.gpu
def test_abi_rejects_bad_arguments_loudly():
    """Error behaviour of the C ABI on the device: every misuse returns non-zero with a message (raised as
    NativeError by the ctypes layer) instead of computing something else."""
    from crowdmod_ddpm_4d_amd import native
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    from crowdmod_ddpm_4d_amd.unet import UNet
    C_ = 3
    net = UNet(input_channels=C_, output_channels=C_, num_res_blocks=1, base_channels=8, base_channels_multiples=(1, 2, 4),
               apply_attention=(False, False, True), max_batch=2)
    net.load_state_dict(spec.init_params(narrow_cfg(C_), SEED_W))
    past, fut = synth_inputs(2, C_, NARROW["H"], NARROW["W"], NARROW["P"], NARROW["F"], "abi")
    out = net(fut, np.array([3, 5]), past)
    assert np.isfinite(out).all()
    L, h = native.lib(), net._handle
    with pytest.raises(native.NativeError, match="timestep"):
        net(fut, np.array([3, 1000]), past)                       # the table has 1000 rows (embeddings.py:7)
    d = native.DeviceBuffer(fut.nbytes)
    with pytest.raises(native.NativeError, match="batch"):
        native.check(L.cm_unet_forward(h, d.ptr, d.ptr, d.ptr, d.ptr, 3, None))   # > max_batch
    with pytest.raises(native.NativeError, match="null"):
        native.check(L.cm_unet_forward(h, None, d.ptr, d.ptr, d.ptr, 2, None))
    s = DDPM(timesteps=50, scale=0.5)
    o = native.cm_sample_opts()
    o.sampler = native.SAMPLER_FM_EULER
    o.fm_steps, o.fm_time_max_pos = 4, 2000
    with pytest.raises(native.NativeError, match="flow-matching"):
        native.check(L.cm_sample_loop(h, s._handle, d.ptr, None, None, C.byref(o), d.ptr, None, 2, None))
    with pytest.raises(native.NativeError, match="cm_train_init"):
        native.check(L.cm_train_apply(h, None))
    with pytest.raises(ValueError):
        net(fut[:, :2], np.array([1, 2]), past)                   # channel mismatch is caught before the call


@pytest.mark.gpu
def test_graph_replay_is_bit_identical_to_eager_launches(monkeypatch):
    """use_graph = 1: one captured step replayed through a hipGraph (per-step scalars in a device table) must give
    exactly the eager loop's result -- DDPM with injected noise + history, DDIM, device-drawn noise."""
    from crowdmod_ddpm_4d_amd.config import AttrDict
    from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    C_, B, T = 3, 2, 20
    cfg = AttrDict({
        "MACROPROPS": {"ROWS": NARROW["H"], "COLS": NARROW["W"]},
        "DATASET": {"PAST_LEN": NARROW["P"], "FUTURE_LEN": NARROW["F"], "BATCH_SIZE": B},
        "MODEL": {"DDPM": {"TIMESTEPS": T, "SCALE": 0.5, "SIGMA": 0.0, "GUIDANCE": "Sparsity", "LAMBDA_GUIDANCE": 0.05, "UNET": {
            "CONDITION": "Past", "NUM_RES_BLOCKS": 1, "BASE_CH": 8, "BASE_CH_MULT": [1, 2, 4],
            "APPLY_ATTENTION": [False, False, True, False], "DROPOUT_RATE": 0.1, "TIME_EMB_MULT": 4}}}})
    past, fut = synth_inputs(B, C_, NARROW["H"], NARROW["W"], NARROW["P"], NARROW["F"], "graph")
    x_T = prng.normal(7, "graph/xT", fut.size).reshape(fut.shape)
    noise = prng.normal(7, "graph/z", T * fut.size).reshape((T,) + fut.shape)
    out = {}
    for mode in ("eager", "graph"):
        if mode == "graph":
            monkeypatch.setenv("CM_USE_GRAPH", "1")
        else:
            monkeypatch.delenv("CM_USE_GRAPH", raising=False)
        m = DDPM_model(cfg, "DDPM-UNet", C_)
        m.denoiser.load_state_dict(spec.init_params(narrow_cfg(C_), SEED_W))
        s = DDPM(timesteps=T, scale=0.5)
        x, hist = m._generate_ddpm(past, s, B, history=True, x_T=x_T, noise=noise)
        xd, _ = m._generate_ddim(past, np.arange(0, T - 1, 3), s, B, x_T=x_T, noise=noise[:7])
        xr, _ = m._generate_ddpm(past, s, B)          # device Philox noise
        out[mode] = (x, np.stack(hist), xd, xr)
    for a, b in zip(out["eager"], out["graph"]):
        assert np.isfinite(a).all() and np.array_equal(a, b)
