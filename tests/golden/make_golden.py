#!/usr/bin/env python3
"""Generate the golden parity fixtures by IMPORTING the reference's own modules.

Runs only in the build container, where /root/reference exists; the reference
never travels to the GPU box -- only the small .npz vectors written here do.

    python tests/golden/make_golden.py [--only schedule,ops,fwd,loop,train]

What is captured:
  schedule.npz      six [T] fp32 tables of ForwardSampler for three (T, scale)
  ops_c{3,4}.npz    per-module inputs/outputs of a narrow UNet (BASE_CH=8)
  fwd.npz           full-width UNet outputs on the ATC / CR-120 / 2x grids
  loop.npz          _generate_ddpm / _generate_ddim results with injected noise
  train.npz         loss + per-tensor gradient norms of one training step
  metrics.npz       PSNR / masked PSNR / relative density / TV tables of the reference's MetricsGenerator
  fm.npz            flow matching: Euler sampling + one training step per probability path
  loop_grids.npz    (--only loop_grids) _generate_ddpm T=50 on the HERMES-CR-120 (28x24) and 2x ATC (24x72) grids,
                    _generate_ddim divider 50 on CR-120
  train_full_cr120.npz  (--only train_grids) full-width training step on the CR-120 grid
  fwd_geoms.npz     (--only fwd_geoms) the reference UNet's forward on the ETH-UCY 8x12 and ATC_medium (16 frames, base 64) geometries
  energy.npz        (--only energy) models/guidance.py compute_energy on synthetic sequences
  motion_feat.npz   (--only motion_feat) utils/metrics/motionFeatureExtractor.py vectors + MF_MSE / MF_BHATT tables

Weights and inputs are NOT stored: both sides regenerate them bit-identically
from the integer PRNG (crowdmod-ddpm-4d_amd/prng.py, spec.init_params).

Import notes: models/backbones/* and models/diffusion/forward.py import with
torch alone.  models/diffusion/ddpm.py (home of DDPM.step/_generate_ddpm) also
imports logging/plot/metrics packages that are absent here (wandb,
torchmetrics, skimage, imageio, easydict, torchvision); none of them is touched
by the arithmetic of the path, so empty placeholder modules are registered for
them before the import.  No reference file is modified or copied.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from crowdmod_ddpm_4d_amd import prng, spec  # noqa: E402

SEED_W, SEED_X = 42, 7


# --------------------------------------------------------------------------- #
def _placeholders():
    class AttrDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, AttrDict):
                v = AttrDict(v)
            super().__setitem__(k, v)

        __setattr__ = __setitem__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def update(self, d=None, **kw):
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

    for name in ("wandb", "torchmetrics", "skimage", "skimage.metrics", "imageio", "imageio.v2",
                 "easydict", "torchvision"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchmetrics"].MeanMetric = type("MeanMetric", (), {})
    sys.modules["skimage.metrics"].structural_similarity = None
    sys.modules["skimage"].metrics = sys.modules["skimage.metrics"]
    sys.modules["easydict"].EasyDict = AttrDict
    return AttrDict


def ref_unet(cfg: spec.UNetConfig, params):
    from models.backbones.unet import UNet
    net = UNet(cfg.input_channels, cfg.output_channels, cfg.num_res_blocks, cfg.base_channels,
               list(cfg.base_channels_multiples), list(cfg.apply_attention), cfg.dropout_rate,
               cfg.time_multiple, cfg.condition)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()}, strict=True)
    return net.eval()


def synth_inputs(B, C, H, W, P, F, tag):
    past = prng.normal(SEED_X, f"past/{tag}", B * C * H * W * P).reshape(B, C, H, W, P)
    fut = prng.normal(SEED_X, f"future/{tag}", B * C * H * W * F).reshape(B, C, H, W, F)
    return past, fut


# --------------------------------------------------------------------------- #
def gen_schedule(out):
    from models.diffusion.forward import ForwardSampler
    d = {}
    for T, scale in ((1000, 0.5), (50, 0.5), (1000, 1.0)):
        fs = ForwardSampler(timesteps=T, scale=scale)
        for name in ("beta", "alpha", "alpha_bar", "sqrt_alpha_bar", "one_by_sqrt_alpha",
                     "sqrt_one_minus_alpha_bar"):
            d[f"T{T}_s{scale}/{name}"] = getattr(fs, name).numpy()
    # q-sample with injected noise: forward.py:29-36 via monkey-patched randn_like
    fs = ForwardSampler(timesteps=1000, scale=0.5)
    x0 = torch.from_numpy(prng.normal(SEED_X, "qs/x0", 4 * 3 * 4 * 8 * 3).reshape(4, 3, 4, 8, 3))
    eps = torch.from_numpy(prng.normal(SEED_X, "qs/eps", 4 * 3 * 4 * 8 * 3).reshape(4, 3, 4, 8, 3))
    t = torch.tensor([0, 17, 500, 999])
    orig = torch.randn_like
    torch.randn_like = lambda x, **kw: eps.clone()
    try:
        xt, e = fs(x0, t)
    finally:
        torch.randn_like = orig
    assert torch.equal(e, eps)
    d["qsample/t"] = t.numpy()
    d["qsample/xt"] = xt.numpy()
    np.savez_compressed(os.path.join(out, "schedule.npz"), **d)
    print("schedule.npz", len(d))


NARROW = dict(H=4, W=8, P=5, F=3, B=2)


def narrow_cfg(C):
    return spec.UNetConfig(C, C, 1, 8, (1, 2, 4), (False, False, True, False), 0.1, 4, "Past")


def gen_ops(out):
    from models.backbones import layers as RL
    import torch.nn as nn
    for C in (3, 4):
        cfg = narrow_cfg(C)
        params = spec.init_params(cfg, SEED_W)
        net = ref_unet(cfg, params)
        g = NARROW
        past, fut = synth_inputs(g["B"], C, g["H"], g["W"], g["P"], g["F"], f"narrow{C}")
        t = np.array([999, 3], dtype=np.int64)
        cap = {}
        with_inputs = {"encoder_blocks.0.conv_1", "encoder_blocks.1.downsample", "encoder_blocks.2.match_input",
                       "encoder_blocks.0.normalize_1", "decoder_blocks.2", "encoder_blocks.4.attention.mhsa",
                       "encoder_blocks.4.attention", "encoder_blocks.0", "encoder_blocks.4", "decoder_blocks.0",
                       "decoder_blocks.3", "final"}
        kinds = (nn.Conv3d, nn.GroupNorm, nn.MultiheadAttention, RL.ResnetBlock, RL.AttentionBlock,
                 RL.DownSample, RL.UpSample, nn.Sequential, nn.Linear)

        def hook(name):
            def fn(mod, inp, outp):
                o = outp[0] if isinstance(outp, tuple) else outp
                cap[f"{name}/out"] = o.detach().numpy().copy()
                if name in with_inputs:
                    cap[f"{name}/in"] = inp[0].detach().numpy().copy()
                    if isinstance(mod, RL.ResnetBlock):
                        cap[f"{name}/temb"] = inp[1].detach().numpy().copy()
            return fn

        for name, mod in net.named_modules():
            if name and isinstance(mod, kinds):
                mod.register_forward_hook(hook(name))
        with torch.inference_mode():
            y = net(torch.from_numpy(fut), torch.from_numpy(t), torch.from_numpy(past))
        cap["t"] = t
        cap["out"] = y.numpy()
        np.savez_compressed(os.path.join(out, f"ops_c{C}.npz"), **cap)
        print(f"ops_c{C}.npz", len(cap), sum(v.nbytes for v in cap.values()) // 1024, "KiB")


FULL_GRIDS = {"atc": (12, 36), "cr120": (28, 24), "atc2x": (24, 72)}


def full_cfg(C):
    return spec.UNetConfig(C, C, 1, 32, (1, 2, 4), (False, False, True, False), 0.1, 4, "Past")


def gen_fwd(out):
    d = {}
    for C in (3, 4):
        cfg = full_cfg(C)
        net = ref_unet(cfg, spec.init_params(cfg, SEED_W))
        for gname, (H, W) in FULL_GRIDS.items():
            if C == 4 and gname != "atc":
                continue
            past, fut = synth_inputs(2, C, H, W, 5, 3, f"full/{gname}/c{C}")
            t = np.array([999, 17], dtype=np.int64)
            with torch.inference_mode():
                y = net(torch.from_numpy(fut), torch.from_numpy(t), torch.from_numpy(past))
            d[f"{gname}_c{C}/t"] = t
            d[f"{gname}_c{C}/out"] = y.numpy()
            print("fwd", gname, C, float(y.abs().max()))
    np.savez_compressed(os.path.join(out, "fwd.npz"), **d)


# the reference's own forward on two of its OTHER shipped geometries (round-3 verdict item 5): ETH-UCY 8x12
# (config/ETHUCY_ddpm.yml:9-10,26-27,38-43) and ATC_medium 12x36 with 8 + 8 frames, BASE_CH 64 (config/ATC_medium.yml:9-10,26-27,38-43)
GEOMS = {"ethucy": (8, 12, 5, 3, 32, (False, False, True, False), 3),
         "atc_medium": (12, 36, 8, 8, 64, (False, False, True), 4)}


def gen_fwd_geoms(out):
    d = {}
    for name, (H, W, P, F, base, att, C) in GEOMS.items():
        cfg = spec.UNetConfig(C, C, 1, base, (1, 2, 4), att, 0.1, 4, "Past")
        net = ref_unet(cfg, spec.init_params(cfg, SEED_W))
        past, fut = synth_inputs(2, C, H, W, P, F, f"geom/{name}")
        t = np.array([999, 17], dtype=np.int64)
        with torch.inference_mode():
            y = net(torch.from_numpy(fut), torch.from_numpy(t), torch.from_numpy(past))
        d[f"{name}/t"] = t
        d[f"{name}/out"] = y.numpy()
        print("fwd_geoms", name, y.shape, float(y.abs().max()))
    np.savez_compressed(os.path.join(out, "fwd_geoms.npz"), **d)


def loop_noise(tag, B, per, t):
    return prng.normal_per_sample(SEED_X, f"z/{tag}", np.arange(B), per, step=t)


def gen_loop(out, grids_only=False):
    AttrDict = _placeholders()
    import yaml
    from models.diffusion import ddpm as RD
    cfg_yaml = AttrDict(yaml.safe_load(open(os.path.join(REF, "config", "ATC.yml"))))
    C, P, F, B = 3, 5, 3, 2
    ucfg = full_cfg(C)
    params = spec.init_params(ucfg, SEED_W)
    d = {}

    def run(tag, T, sampler, guidance="None", lam=0.0, divider=None, sigma=None, keep=(), grid=(12, 36)):
        H, W = grid
        cfg = AttrDict(yaml.safe_load(open(os.path.join(REF, "config", "ATC.yml"))))
        cfg.MACROPROPS.ROWS, cfg.MACROPROPS.COLS = H, W
        cfg.MODEL.DDPM.TIMESTEPS = T
        cfg.MODEL.DDPM.GUIDANCE = guidance
        cfg.MODEL.DDPM.LAMBDA_GUIDANCE = lam
        if sigma is not None:
            cfg.MODEL.DDPM.SIGMA = sigma
        model = RD.DDPM_model(cfg, "DDPM-UNet", C)
        model.denoiser.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
        sampler_obj = RD.DDPM(timesteps=T, scale=cfg.MODEL.DDPM.SCALE)
        per = C * H * W * F
        past = torch.from_numpy(prng.normal(SEED_X, f"past/loop/{tag}", B * C * H * W * P).reshape(B, C, H, W, P))
        x_T = prng.normal_per_sample(SEED_X, f"xT/{tag}", np.arange(B), per).reshape(B, C, H, W, F)
        # The reference draws x_T with torch.randn and z_t with torch.randn_like
        # (ddpm.py:27,211,264).  Inject ours by patching those two entry points
        # while the reference's own loop runs; the visited t is recovered from
        # the call order.
        if sampler == "DDPM":
            order = [t for t in reversed(range(T)) if t > 0]
        else:
            taus = np.arange(0, T - 1, divider)
            order = list(reversed(list(taus)))
        calls = {"n": 0}
        hist = {}

        def fake_randn(*a, **kw):
            return torch.from_numpy(x_T.copy())

        def fake_randn_like(x, **kw):
            t = int(order[calls["n"]])
            calls["n"] += 1
            return torch.from_numpy(loop_noise(tag, B, per, t).reshape(x.shape))

        o1, o2 = torch.randn, torch.randn_like
        torch.randn, torch.randn_like = fake_randn, fake_randn_like
        try:
            if sampler == "DDPM":
                x, h = model._generate_ddpm(past, sampler_obj, B, history=bool(keep))
                if keep:   # history = [x_T, x after t=T-1, ..., x after t=0]
                    for t in keep:
                        hist[t] = h[1 + (T - 1 - t)].numpy().copy()
            else:
                x, _ = model._generate_ddim(past, taus, sampler_obj, B)
        finally:
            torch.randn, torch.randn_like = o1, o2
        assert calls["n"] == len(order), (calls, len(order))
        d[f"{tag}/x0"] = x.numpy()
        for t, v in hist.items():
            d[f"{tag}/x_after_t{t}"] = v
        print("loop", tag, float(x.abs().max()))

    if grids_only:
        # round 2: the reference's own loop on the other two grids of BASELINE.json (configs[3] HERMES-CR-120 28x24,
        # configs[4] 2x ATC 24x72) -- their own file, so that loop.npz keeps regenerating bit for bit
        run("cr120_ddpm50", 50, "DDPM", grid=FULL_GRIDS["cr120"])
        run("atc2x_ddpm50", 50, "DDPM", grid=FULL_GRIDS["atc2x"])
        run("cr120_ddim1000_div50", 1000, "DDIM", divider=50, sigma=0.001, grid=FULL_GRIDS["cr120"])
        np.savez_compressed(os.path.join(out, "loop_grids.npz"), **d)
        return
    run("ddpm50", 50, "DDPM")
    run("ddpm20_sparsity", 20, "DDPM", guidance="Sparsity", lam=0.004)
    run("ddim1000_div100", 1000, "DDIM", divider=100, sigma=0.001)
    run("ddim1000_div100_sparsity", 1000, "DDIM", divider=100, sigma=0.001, guidance="Sparsity", lam=0.004)
    run("ddpm1000", 1000, "DDPM", keep=(999, 900, 500, 0))
    np.savez_compressed(os.path.join(out, "loop.npz"), **d)


def gen_train(out):
    """One fp32 training step of the narrow model: ddpm.py:111-121 with t, eps and
    the Dropout3d masks injected (autocast is a no-op on CPU)."""
    import torch.nn as nn
    from models.backbones import layers as RL
    from models.diffusion.forward import ForwardSampler
    C = 3
    cfg = narrow_cfg(C)
    params = spec.init_params(cfg, SEED_W)
    net = ref_unet(cfg, params).train()
    g = NARROW
    B = 4
    past, fut = synth_inputs(B, C, g["H"], g["W"], g["P"], g["F"], "train")
    eps = prng.normal(SEED_X, "train/eps", fut.size).reshape(fut.shape)
    t = np.array([0, 250, 731, 999], dtype=np.int64)
    p = cfg.dropout_rate

    class FixedDrop(nn.Module):
        def __init__(self, mask):
            super().__init__()
            self.mask = mask

        def forward(self, x):
            return x * self.mask[:, :, None, None, None]

    masks = {}
    for name, mod in net.named_modules():
        if isinstance(mod, RL.ResnetBlock):
            u = prng.uniform_pm1(SEED_X, f"drop/{name}", B * mod.out_channels).reshape(B, mod.out_channels)
            keep = ((u * 0.5 + 0.5) >= p).astype(np.float32) / np.float32(1.0 - p)
            masks[name] = keep
            mod.dropout = FixedDrop(torch.from_numpy(keep))
    fs = ForwardSampler(timesteps=1000, scale=0.5)
    o = torch.randn_like
    torch.randn_like = lambda x, **kw: torch.from_numpy(eps.copy())
    try:
        xt, e = fs(torch.from_numpy(fut), torch.from_numpy(t))
    finally:
        torch.randn_like = o
    pred = net(xt, torch.from_numpy(t), torch.from_numpy(past))
    loss = torch.nn.functional.mse_loss(pred, e)
    loss.backward()
    d = {"t": t, "loss": np.float32(loss.item()), "pred": pred.detach().numpy()}
    for name, prm in net.named_parameters():
        if prm.grad is not None:
            d[f"gnorm/{name}"] = np.float32(prm.grad.norm().item())
    for k in ("first.weight", "final.2.weight", "encoder_blocks.4.attention.mhsa.in_proj_weight",
              "encoder_blocks.0.conv_1.weight", "decoder_blocks.7.normalize_1.weight",
              "time_embeddings.time_blocks.1.weight"):
        d[f"grad/{k}"] = dict(net.named_parameters())[k].grad.numpy()
    # optimizer.step() of ddpm.py:53-56,142-144 (config/ATC.yml solver: lr 5e-5, betas (0.5, 0.999),
    # weight_decay 0.003), two steps so that the moment buffers and the bias correction are exercised:
    # the second step re-runs the same batch / masks on the updated weights
    opt = torch.optim.Adam(net.parameters(), lr=5e-5, betas=(0.5, 0.999), weight_decay=0.003)
    opt.step()
    post = ("first.weight", "final.2.weight", "encoder_blocks.4.attention.mhsa.in_proj_weight",
            "encoder_blocks.0.conv_1.weight", "decoder_blocks.7.normalize_1.weight", "decoder_blocks.5.upsample.1.weight",
            "encoder_blocks.1.downsample.weight", "time_embeddings.time_blocks.3.weight", "bottleneck_blocks.0.dense_1.bias")
    sd = net.state_dict()
    for k in post:
        d[f"post1/{k}"] = sd[k].detach().numpy().copy()
    opt.zero_grad(set_to_none=True)
    pred2 = net(xt, torch.from_numpy(t), torch.from_numpy(past))
    loss2 = torch.nn.functional.mse_loss(pred2, e)
    loss2.backward()
    opt.step()
    d["loss2"] = np.float32(loss2.item())
    sd = net.state_dict()
    for k in post:
        d[f"post2/{k}"] = sd[k].detach().numpy().copy()
    np.savez_compressed(os.path.join(out, "train.npz"), **d)
    print("train loss", float(loss), "second step", float(loss2))


def gen_train_full(out, grid="atc"):
    """Training step of the FULL-width model (base 32, ATC grid 12x36, B = 2): loss and the gradient norm of
    every trainable tensor -- exercises the 32-channel-chunk kernels, the K-split quarter-resolution layers and
    the parity-form upsample convs of the backward pass, which the narrow model does not reach.
    grid="cr120" (round 2): the same on the HERMES-CR-120 grid 28x24 -> train_full_cr120.npz."""
    import torch.nn as nn
    from models.backbones import layers as RL
    from models.diffusion.forward import ForwardSampler
    C, B = 3, 2
    H, W = FULL_GRIDS[grid]
    P, F = 5, 3
    cfg = full_cfg(C)
    params = spec.init_params(cfg, SEED_W)
    net = ref_unet(cfg, params).train()
    sfx = "" if grid == "atc" else "/" + grid
    past, fut = synth_inputs(B, C, H, W, P, F, "trainfull" + sfx)
    eps = prng.normal(SEED_X, "trainfull/eps" + sfx, fut.size).reshape(fut.shape)
    t = np.array([17, 803], dtype=np.int64)

    class FixedDrop(nn.Module):
        def __init__(self, mask):
            super().__init__()
            self.mask = mask

        def forward(self, x):
            return x * self.mask[:, :, None, None, None]

    for name, mod in net.named_modules():
        if isinstance(mod, RL.ResnetBlock):
            u = prng.uniform_pm1(SEED_X, f"dropfull/{name}", B * mod.out_channels).reshape(B, mod.out_channels)
            keep = ((u * 0.5 + 0.5) >= 0.1).astype(np.float32) / np.float32(0.9)
            mod.dropout = FixedDrop(torch.from_numpy(keep))
    fs = ForwardSampler(timesteps=1000, scale=0.5)
    o = torch.randn_like
    torch.randn_like = lambda x, **kw: torch.from_numpy(eps.copy())
    try:
        xt, e = fs(torch.from_numpy(fut), torch.from_numpy(t))
    finally:
        torch.randn_like = o
    pred = net(xt, torch.from_numpy(t), torch.from_numpy(past))
    loss = torch.nn.functional.mse_loss(pred, e)
    loss.backward()
    d = {"t": t, "loss": np.float32(loss.item())}
    for name, prm in net.named_parameters():
        if prm.grad is not None:
            d[f"gnorm/{name}"] = np.float32(prm.grad.norm().item())
    for k in ("decoder_blocks.5.upsample.1.weight", "bottleneck_blocks.0.conv_1.weight"):
        g = dict(net.named_parameters())[k].grad.numpy()
        d[f"gslice/{k}"] = g[:4, :4].copy()      # a 4 x 4 x 3 x 3 x 3 corner of two large gradients
    np.savez_compressed(os.path.join(out, "train_full.npz" if grid == "atc" else f"train_full_{grid}.npz"), **d)
    print("train_full", grid, "loss", float(loss.item()), "tensors", len(d) - 2)


def gen_fm(out):
    """Flow matching on the UNet backbone (models/flow_matching/flow_matching.py): the reference's own
    FM_model.sampling_with_euler with x_0 injected (torch.randn patched) on the narrow model, and one
    training step of _train_one_epoch_fm's body for the Linear and the Conic path with x_0, t and the
    Dropout3d masks injected (loss + gradient norms)."""
    AttrDict = _placeholders()
    import torch.nn as nn
    from models.backbones import layers as RL
    from models.flow_matching import flow_matching as RF
    C, B = 3, 2
    g = NARROW
    ucfg = narrow_cfg(C)
    params = spec.init_params(ucfg, SEED_W)
    cfg = AttrDict({
        "MACROPROPS": {"ROWS": g["H"], "COLS": g["W"]},
        "DATASET": {"PAST_LEN": g["P"], "FUTURE_LEN": g["F"], "BATCH_SIZE": B},
        "MODEL": {"NSAMPLES4PLOTS": B, "FM": {
            "TIME_MAX_POS": 1000, "CHECKPOINTS_TO_KEEP": 1, "W_TYPE": "Linear", "INTEGRATOR": "Euler",
            "INTEGRATOR_STEPS": {"EULER": 8, "HEUN": 4},
            "UNET": {"CONDITION": "Past", "NUM_RES_BLOCKS": 1, "BASE_CH": 8, "BASE_CH_MULT": [1, 2, 4],
                     "APPLY_ATTENTION": [False, False, True, False], "DROPOUT_RATE": 0.1, "TIME_EMB_MULT": 4,
                     "TRAIN": {"EPOCHS": 1, "SOLVER": {"LR": 1e-4, "BETAS": [0.5, 0.999], "WEIGHT_DECAY": 0.001,
                               "SCHEDULER": {"FACTOR": 0.5, "PATIENCE": 10, "MIN_LR": 1e-6}}}}}}})
    model = RF.FM_model(cfg, "FM-UNet", C)
    model.u_predictor.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    d = {}
    past, fut = synth_inputs(B, C, g["H"], g["W"], g["P"], g["F"], "fm")
    x0 = prng.normal(SEED_X, "fm/x0", fut.size).reshape(fut.shape)
    o = torch.randn
    torch.randn = lambda *a, **kw: torch.from_numpy(x0.copy())
    try:
        d["euler8"] = model.sampling_with_euler(torch.from_numpy(past), B).numpy()
    finally:
        torch.randn = o
    # training step body (flow_matching.py:128-146) with injected x0 / t / masks
    t = np.array([0.137, 0.862], dtype=np.float32)
    d["t"] = t

    class FixedDrop(nn.Module):
        def __init__(self, mask):
            super().__init__()
            self.mask = mask

        def forward(self, x):
            return x * self.mask[:, :, None, None, None]

    net = model.u_predictor.train()
    for name, mod in net.named_modules():
        if isinstance(mod, RL.ResnetBlock):
            u = prng.uniform_pm1(SEED_X, f"drop/{name}", B * mod.out_channels).reshape(B, mod.out_channels)
            keep = ((u * 0.5 + 0.5) >= 0.1).astype(np.float32) / np.float32(0.9)
            mod.dropout = FixedDrop(torch.from_numpy(keep))
    for wtype in ("Linear", "Conic"):
        net.zero_grad(set_to_none=True)
        tt = torch.from_numpy(t).view(-1, 1, 1, 1, 1)
        xt, u_target = model.w_type_fns[wtype](torch.from_numpy(x0), torch.from_numpy(fut), tt)
        u_pred = net(xt, (tt * cfg.MODEL.FM.TIME_MAX_POS).long().view(-1), torch.from_numpy(past))
        loss = ((u_target - u_pred) ** 2).mean()
        loss.backward()
        d[f"{wtype}/loss"] = np.float32(loss.item())
        d[f"{wtype}/pred"] = u_pred.detach().numpy()
        for name, prm in net.named_parameters():
            if prm.grad is not None and (name.endswith("conv_2.weight") or name.startswith("final") or name.startswith("first")):
                d[f"{wtype}/gnorm/{name}"] = np.float32(prm.grad.norm().item())
    np.savez_compressed(os.path.join(out, "fm.npz"), **d)
    print("fm euler |x|max", float(np.abs(d["euler8"]).max()), "losses", float(d["Linear/loss"]), float(d["Conic/loss"]))


def gen_metrics(out):
    """The reduction-type metrics of the reference's MetricsGenerator (utils/metrics/metricsGenerator.py):
    PSNR / masked PSNR (+ MAX, + over time), relative density error (+ MIN), TV over time, on 8 synthetic
    (prediction, ground truth) sequences = 2 pasts x 4 repeats."""
    AttrDict = _placeholders()
    from utils.metrics.metricsGenerator import MetricsGenerator
    N, C, H, W, F, chunk = 8, 3, 12, 36, 3, 4
    gt = prng.normal(SEED_X, "metrics/gt", N * C * H * W * F).reshape(N, C, H, W, F)
    gt[:, 0] = np.maximum(gt[:, 0], 0.0)          # density: non-negative with empty cells (the mask matters)
    gt[0, 0, :, :, 2] = 0.0                        # one frame with an empty mask (replicated over its repeats)
    pred = (gt + 0.3 * prng.normal(SEED_X, "metrics/noise", gt.size).reshape(gt.shape)).astype(np.float32)
    for i in range(0, N, chunk):                   # repeats of one past share the ground truth
        gt[i:i + chunk] = gt[i]
    gt = gt.astype(np.float32)
    mg = MetricsGenerator([torch.from_numpy(p) for p in pred], [torch.from_numpy(g) for g in gt],
                          AttrDict({"MPROPS_COUNT": 3}), None)
    eps = 1e-8
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mg.compute_psnr_metric(chunk, eps, masked_flag=False)
        mg.compute_psnr_metric(chunk, eps, masked_flag=True)
    mg.compute_re_density_metric(chunk, eps)
    mg.compute_tv_metric()
    d = {"ranges": np.array([mg.rho_range, mg.vx_range, mg.vy_range]), "chunk": np.int64(chunk), "eps": np.float64(eps)}
    for k in ("PSNR", "MAX_PSNR", "PSNR_OVER_TIME", "MAX_PSNR_OVER_TIME", "MASK_PSNR", "MAX_MASK_PSNR", "MASK_PSNR_OVER_TIME",
              "MAX_MASK_PSNR_OVER_TIME", "RE_DENSITY", "MIN_RE_DENSITY", "TV_OVER_TIME"):
        d[k] = np.asarray(mg.data_dict[k], dtype=np.float64)
    np.savez_compressed(os.path.join(out, "metrics.npz"), **d)
    print("metrics psnr", d["PSNR"][0], "masked nan count", int(np.isnan(d["MASK_PSNR_OVER_TIME"]).sum()))


def gen_energy(out):
    """models/guidance.py:10-42 compute_energy (torch-only import) on synthetic [B,3,H,W,L] tensors, with the metric's
    delta_t = delta_l = 1 (metricsGenerator.py:278-279) and the guidance default (0.5, 1.0)."""
    from models.guidance import compute_energy
    B, C, H, W, L = 6, 3, 12, 36, 3
    x = prng.normal(SEED_X, "energy/x", B * C * H * W * L).reshape(B, C, H, W, L).astype(np.float32)
    x[:, 0] = np.maximum(x[:, 0], 0.0)
    t = torch.from_numpy(x)
    d = {"e11": compute_energy(t, delta_t=1, delta_l=1).numpy().astype(np.float32),
         "e_default": compute_energy(t).numpy().astype(np.float32)}
    np.savez_compressed(os.path.join(out, "energy.npz"), **d)
    print("energy", d["e11"][:3])


def gen_motion_feat(out):
    """utils/metrics/motionFeatureExtractor.py + MetricsGenerator.compute_motion_feature_metrics (metricsGenerator.py:240-258)
    on synthetic velocity fields: the two motion-feature vectors of every sequence and the MF_MSE / MF_BHATT tables, for
    the ATC setting (f 1, k 4, gamma 0.5) and a ragged one (f 2, k 5, gamma 2: volumes cut off at the far edges).  The
    reference's histogram plots (matplotlib, drawn for a random 5 % of the volumes) are switched off."""
    AttrDict = _placeholders()
    import utils.metrics.motionFeatureExtractor as MFE
    import utils.metrics.metricsGenerator as MG
    for mod in (MFE, MG):
        for fn in ("plot_motion_feat_hist2D", "plot_motion_feat_hist1D"):
            if hasattr(mod, fn):
                setattr(mod, fn, lambda *a, **k: None)
    N, C, H, W, F = 6, 3, 12, 36, 3
    gt = prng.normal(SEED_X, "mf/gt", N * C * H * W * F).reshape(N, C, H, W, F)
    gt[:, 1:3, :3, :, :] = 0.0                      # motionless cells (first magnitude bin, angle 0)
    gt[:, 1:3, 3:5, :, :] = gt[:, 1:3, 3:5, :, :1]  # cells whose velocity never changes (zero range under the min-max scaling)
    gt[0, 1, 6, :, :] = -np.abs(gt[0, 1, 6, :, :])  # angles of exactly +pi (v_y = +0) and -pi ... (v_y = -0)
    gt[0, 2, 6, :18, :] = 0.0
    gt[0, 2, 6, 18:, :] = -0.0
    pred = gt + 0.4 * prng.normal(SEED_X, "mf/noise", gt.size).reshape(gt.shape)
    pred[:, 1:3, :2] = 0.0
    gt, pred = gt.astype(np.float32), pred.astype(np.float32)
    d = {"gt": gt, "pred": pred}
    for tag, f, k, gamma in (("atc", 1, 4, 0.5), ("ragged", 2, 5, 2.0)):
        params = AttrDict({"MPROPS_COUNT": 3, "MOTION_FEATURE": {"f": f, "k": k, "GAMMA": gamma}})
        mg = MG.MetricsGenerator([torch.from_numpy(p) for p in pred], [torch.from_numpy(g) for g in gt], params, None)
        mg.compute_motion_feature_metrics(True, True)
        ep = MFE.MotionFeatureExtractor(mg.pred_seq_list, f=f, k=k, gamma=gamma)
        eg = MFE.MotionFeatureExtractor(mg.gt_seq_list, f=f, k=k, gamma=gamma)
        p2, g2 = MFE.get_motion_feature_2D_hist(ep, eg)
        p1, g1 = MFE.get_motion_feature_1D_hist(ep, eg)
        d.update({f"{tag}/fkg": np.array([f, k, gamma]), f"{tag}/p2": p2, f"{tag}/g2": g2, f"{tag}/p1": p1, f"{tag}/g1": g1,
                  f"{tag}/mag": ep.mag_rho_transf, f"{tag}/ang": ep.angle_phi})
        for name in ("MF_MSE", "MF_BHATT_DIST", "MF_BHATT_COEF"):
            d[f"{tag}/{name}"] = np.asarray(mg.data_dict[name], dtype=np.float64)
        print("motion features", tag, p2.shape, p1.shape, d[f"{tag}/MF_BHATT_COEF"][0])
    np.savez_compressed(os.path.join(out, "motion_feat.npz"), **d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="schedule,ops,fwd,loop,train,train_full,fm,metrics,energy,motion_feat")
    ap.add_argument("--out", default=HERE)
    a = ap.parse_args()
    torch.manual_seed(0)
    todo = set(a.only.split(","))
    if "schedule" in todo:
        gen_schedule(a.out)
    if "ops" in todo:
        gen_ops(a.out)
    if "fwd" in todo:
        gen_fwd(a.out)
    if "fwd_geoms" in todo:
        gen_fwd_geoms(a.out)
    if "train" in todo:
        gen_train(a.out)
    if "train_full" in todo:
        gen_train_full(a.out)
    if "fm" in todo:
        gen_fm(a.out)
    if "metrics" in todo:
        gen_metrics(a.out)
    if "energy" in todo:
        gen_energy(a.out)
    if "motion_feat" in todo:
        gen_motion_feat(a.out)
    if "loop" in todo:
        gen_loop(a.out)
    if "train_grids" in todo:
        gen_train_full(a.out, grid="cr120")
    if "loop_grids" in todo:
        gen_loop(a.out, grids_only=True)


if __name__ == "__main__":
    main()
