"""The reference's OTHER shipped geometries through the HIP path (round-3 verdict, item 5): ETH-UCY 8x12, HERMES-BN 28x16,
HERMES-BO 12x24, HERMES-CR-90 12x20 (C = 3, base 32; /root/reference/config/ETHUCY_ddpm.yml:9-10, HERMES-BN.yml:10-11,
HERMES-BO.yml:10-11, HERMES-CR-90.yml:10-11) and ATC_medium (12x36, PAST 8 + FUTURE 8 = 16 frames, BASE_CH 64,
/root/reference/config/ATC_medium.yml:26-41).  Forward and a 20-step reverse loop at B = 2 and at B = 9 (odd, > 8) against the
CPU oracle (oracle/unet_torch.py, itself pinned by the reference's fixtures) <= 1e-4; one training step on ATC_medium.
The ETH-UCY and ATC_medium forwards are ALSO pinned by the reference itself (tests/golden/fwd_geoms.npz, make_golden.py --only
fwd_geoms).  Reference lines: models/backbones/unet.py:124-167, models/diffusion/ddpm.py:206-236,111-121."""
import numpy as np
import pytest

from crowdmod_ddpm_4d_amd import prng, spec
from helpers import GOLDEN, SEED_W, synth_inputs

pytestmark = pytest.mark.gpu

TOL = 1e-4
# name: (H, W, past, future, base, attention flags, channels)
GEOMS = {
    "ethucy": (8, 12, 5, 3, 32, (False, False, True, False), 3),
    "hermes_bn": (28, 16, 5, 3, 32, (False, False, True, False), 3),
    "hermes_bo": (12, 24, 5, 3, 32, (False, False, True, False), 3),
    "hermes_cr90": (12, 20, 5, 3, 32, (False, False, True, False), 3),
    "atc_medium": (12, 36, 8, 8, 64, (False, False, True), 4),
}


def _cfg(name):
    H, W, P, F, base, att, C = GEOMS[name]
    return spec.UNetConfig(C, C, 1, base, (1, 2, 4), att, 0.1, 4, "Past")


def _model(name, T, B):
    from crowdmod_ddpm_4d_amd.config import AttrDict
    from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model
    H, W, P, F, base, att, C = GEOMS[name]
    cfg = AttrDict({
        "MACROPROPS": {"ROWS": H, "COLS": W}, "DATASET": {"PAST_LEN": P, "FUTURE_LEN": F, "BATCH_SIZE": B},
        "MODEL": {"NSAMPLES": B, "NSAMPLES4PLOTS": B, "DDPM": {
            "SAMPLER": "DDPM", "TIMESTEPS": T, "SCALE": 0.5, "SIGMA": 0.001, "DDIM_DIVIDER": 2, "GUIDANCE": "None",
            "LAMBDA_GUIDANCE": 0.0,
            "UNET": {"CONDITION": "Past", "NUM_RES_BLOCKS": 1, "BASE_CH": base, "BASE_CH_MULT": [1, 2, 4],
                     "APPLY_ATTENTION": list(att), "DROPOUT_RATE": 0.1, "TIME_EMB_MULT": 4}}}})
    m = DDPM_model(cfg, "DDPM-UNet", C)
    m.denoiser.load_state_dict(spec.init_params(_cfg(name), SEED_W))
    return m


def _torch_params(name):
    import torch
    from oracle import unet_torch as ot
    torch.set_num_threads(16)
    cfg = _cfg(name)
    return ot, ot.to_torch(spec.init_params(cfg, SEED_W)), spec.make_plan(cfg)


@pytest.mark.parametrize("name", list(GEOMS))
@pytest.mark.parametrize("B", [2, 9])
def test_forward_and_20_step_loop_vs_oracle(name, B):
    import torch
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    H, W, P, F, base, att, C = GEOMS[name]
    ot, Pt, plan = _torch_params(name)
    past, fut = synth_inputs(B, C, H, W, P, F, f"geom/{name}")
    t = (np.arange(B) * 113 + 7) % 1000
    # ---- one forward ----
    m = _model(name, 20, B)
    y = m.denoiser(fut, t, past)
    with torch.no_grad():
        ref = ot.unet_forward(Pt, plan, torch.from_numpy(fut), torch.from_numpy(t).long(), torch.from_numpy(past), None).numpy()
    err = float(np.abs(y - ref).max())
    assert err <= TOL, (name, B, err)
    # ---- the reverse loop, T = 20, injected x_T and noise ----
    T = 20
    per = C * H * W * F
    x_T = prng.normal_per_sample(7, f"geom/{name}/xT", np.arange(B), per).reshape(B, C, H, W, F)
    noise = np.stack([prng.normal_per_sample(7, f"geom/{name}/z", np.arange(B), per, step=s).reshape(B, C, H, W, F)
                      for s in range(T - 1, 0, -1)])
    x, _ = m._generate_ddpm(past, DDPM(timesteps=T, scale=0.5), B, x_T=x_T, noise=noise)
    sched = ot.schedule(T, 0.5)
    with torch.no_grad():
        xr = ot.generate_ddpm(Pt, plan, sched, torch.from_numpy(past), torch.from_numpy(x_T),
                              lambda s: torch.from_numpy(noise[T - 1 - s]), T).numpy()
    errl = float(np.abs(x - xr).max())
    assert errl <= TOL * max(1.0, float(np.abs(xr).max())), (name, B, errl, float(np.abs(xr).max()))


@pytest.mark.parametrize("name", ["ethucy", "atc_medium"])
def test_forward_vs_the_reference_itself(name):
    """Forwards the reference's own UNet produced for these two geometries (make_golden.py --only fwd_geoms)."""
    import os
    g = np.load(os.path.join(GOLDEN, "fwd_geoms.npz"))
    H, W, P, F, base, att, C = GEOMS[name]
    past, fut = synth_inputs(2, C, H, W, P, F, f"geom/{name}")
    m = _model(name, 20, 2)
    y = m.denoiser(fut, g[name + "/t"], past)
    err = float(np.abs(y - g[name + "/out"]).max())
    assert err <= TOL, (name, err)


def test_training_step_on_atc_medium_vs_oracle_autograd():
    """One training step (loss + every parameter gradient) on the 16-frame, base-64 geometry: four z planes at quarter
    resolution -- outside the two-plane kernels -- and 64 / 128 / 256 channels."""
    import torch
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    from crowdmod_ddpm_4d_amd.unet import UNet
    name, B = "atc_medium", 2
    H, W, P, F, base, att, C = GEOMS[name]
    ot, _, plan = _torch_params(name)
    cfg = _cfg(name)
    params = spec.init_params(cfg, SEED_W)
    past, fut = synth_inputs(B, C, H, W, P, F, f"geom/{name}")
    eps = prng.normal(11, "geomtrain/eps", fut.size).reshape(fut.shape)
    t = np.array([3, 777])
    masks = {}
    for blk in plan.res_blocks():
        u = prng.uniform_pm1(11, f"dropgeom/{blk.prefix}", B * blk.cout).reshape(B, blk.cout)
        masks[blk.prefix] = ((u * 0.5 + 0.5) >= 0.1).astype(np.float32) / np.float32(0.9)
    Pg = {k: torch.tensor(np.asarray(v), dtype=torch.float32) for k, v in params.items()}
    for k, v in Pg.items():
        if k != "time_embeddings.time_blocks.0.weight":
            v.requires_grad_(True)
    sched = ot.schedule(1000, scale=0.5)
    tt = torch.tensor(t, dtype=torch.long)
    x0, e = torch.tensor(fut), torch.tensor(eps)
    xt = sched["sqrt_alpha_bar"][tt].view(-1, 1, 1, 1, 1) * x0 + sched["sqrt_one_minus_alpha_bar"][tt].view(-1, 1, 1, 1, 1) * e
    pred = ot.unet_forward(Pg, plan, xt, tt, torch.tensor(past), {k: torch.tensor(v) for k, v in masks.items()})
    loss_ref = torch.nn.functional.mse_loss(pred, e)
    loss_ref.backward()
    net = UNet(input_channels=C, output_channels=C, num_res_blocks=1, base_channels=base, base_channels_multiples=(1, 2, 4),
               apply_attention=att, dropout_rate=0.1, time_multiple=4, condition="Past", max_batch=B)
    net.load_state_dict(params)
    net.ensure(H, W, P, F, B)
    net.train_init(lr=5e-5, betas=(0.5, 0.999), weight_decay=0.003)
    sampler = DDPM(timesteps=1000, scale=0.5)          # (kept alive: its handle owns the device tables the step reads)
    loss = net.train_step(sampler._handle, fut, past, t, eps, drop_masks=masks, apply_update=False)
    loss_ref = loss_ref.detach()
    assert abs(loss - float(loss_ref)) <= 1e-5 * max(1.0, float(loss_ref)), (loss, float(loss_ref))
    n = 0
    for k, v in Pg.items():
        if v.grad is None:
            continue
        gref = v.grad.numpy()
        got = net.grad(k).astype(np.float64).reshape(gref.shape)
        err, scale = float(np.abs(got - gref).max()), float(np.abs(gref).max())
        assert err <= 3e-4 * scale + 2e-8, (k, err, scale)
        n += 1
    assert n >= 160, n
