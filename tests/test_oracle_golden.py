"""Pin the oracle (oracle/unet_numpy.py, oracle/unet_torch.py) against the
golden vectors captured from the reference's own modules.  CPU only."""
import numpy as np
import pytest
import torch

from crowdmod_ddpm_4d_amd import spec
from oracle import unet_numpy as on
from oracle import unet_torch as ot
from helpers import (FULL_GRIDS, NARROW, SEED_W, full_cfg, load, loop_noise, narrow_cfg, synth_inputs)
from crowdmod_ddpm_4d_amd import prng

TOL = 1e-4  # north_star: <= 1e-4 max-abs vs reference per forward


@pytest.mark.parametrize("T,scale", [(1000, 0.5), (50, 0.5), (1000, 1.0)])
def test_schedule_tables_bit_exact(T, scale):
    g = load("schedule.npz")
    s = on.schedule(T, scale)
    for k, v in s.items():
        ref = g[f"T{T}_s{scale}/{k}"]
        if k in ("beta", "alpha", "alpha_bar"):
            assert np.array_equal(v, ref), k
        else:  # sqrt / reciprocal: torch's vectorised sqrt differs by <= 1 ulp on a few entries
            np.testing.assert_allclose(v, ref, rtol=1.2e-7, atol=0, err_msg=k)


def test_q_sample():
    g = load("schedule.npz")
    s = on.schedule(1000, 0.5)
    x0 = prng.normal(7, "qs/x0", 4 * 3 * 4 * 8 * 3).reshape(4, 3, 4, 8, 3)
    eps = prng.normal(7, "qs/eps", 4 * 3 * 4 * 8 * 3).reshape(4, 3, 4, 8, 3)
    xt = on.q_sample(s, x0, g["qsample/t"], eps)
    np.testing.assert_allclose(xt, g["qsample/xt"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("C", [3, 4])
def test_ops_numpy(C):
    g = load(f"ops_c{C}.npz")
    cfg = narrow_cfg(C)
    P = spec.init_params(cfg, SEED_W)
    plan = spec.make_plan(cfg)
    f32 = np.float32

    def close(a, b, what):
        err = np.abs(a - b).max()
        assert err <= 2e-5, (what, err)

    # conv 3x3x3 stride 1 / stride 2 / 1x1x1
    close(on.conv3d(g["encoder_blocks.0.conv_1/in"], P["encoder_blocks.0.conv_1.weight"],
                    P["encoder_blocks.0.conv_1.bias"]), g["encoder_blocks.0.conv_1/out"], "conv s1")
    close(on.conv3d(g["encoder_blocks.1.downsample/in"], P["encoder_blocks.1.downsample.weight"],
                    P["encoder_blocks.1.downsample.bias"], stride=2), g["encoder_blocks.1.downsample/out"], "conv s2")
    close(on.conv3d(g["encoder_blocks.2.match_input/in"], P["encoder_blocks.2.match_input.weight"],
                    P["encoder_blocks.2.match_input.bias"], pad=0), g["encoder_blocks.2.match_input/out"], "conv 1x1")
    # group norm
    close(on.group_norm(g["encoder_blocks.0.normalize_1/in"], P["encoder_blocks.0.normalize_1.weight"],
                        P["encoder_blocks.0.normalize_1.bias"]), g["encoder_blocks.0.normalize_1/out"], "gn")
    # upsample + conv
    up = on.conv3d(on.upsample_nearest2(g["decoder_blocks.2/in"]), P["decoder_blocks.2.upsample.1.weight"],
                   P["decoder_blocks.2.upsample.1.bias"])
    close(up, g["decoder_blocks.2/out"], "upsample")
    # MHA and attention block
    pre = "encoder_blocks.4.attention"
    close(on.mha_self(g[pre + ".mhsa/in"], P[pre + ".mhsa.in_proj_weight"], P[pre + ".mhsa.in_proj_bias"],
                      P[pre + ".mhsa.out_proj.weight"], P[pre + ".mhsa.out_proj.bias"]), g[pre + ".mhsa/out"], "mha")
    close(on.attention_block(g[pre + "/in"], P, pre), g[pre + "/out"], "attn block")
    # time embedding
    close(on.time_embedding(g["t"], P), g["time_embeddings.time_blocks/out"], "time mlp")
    # resnet blocks: plain, 1x1 skip + attention, decoder concat input
    for name in ("encoder_blocks.0", "encoder_blocks.4", "decoder_blocks.0", "decoder_blocks.3"):
        close(on.resnet_block(g[name + "/in"], g[name + "/temb"], P, name), g[name + "/out"], name)
    # whole forward with per-block trace
    geo = NARROW
    past, fut = synth_inputs(geo["B"], C, geo["H"], geo["W"], geo["P"], geo["F"], f"narrow{C}")
    trace = {}
    y = on.unet_forward(P, plan, fut, g["t"], past, trace=trace)
    for k, v in trace.items():
        close(v, g[k + "/out"], k)
    close(y, g["out"], "unet out")
    # torch restatement
    Pt = ot.to_torch(P)
    yt = ot.unet_forward(Pt, plan, torch.from_numpy(fut), torch.from_numpy(g["t"]), torch.from_numpy(past)).numpy()
    close(yt, g["out"], "unet out (torch oracle)")


@pytest.mark.parametrize("key", ["atc_c3", "atc_c4", "cr120_c3", "atc2x_c3"])
def test_full_forward_torch_oracle(key):
    g = load("fwd.npz")
    gname, c = key.split("_c")
    C = int(c)
    H, W = FULL_GRIDS[gname]
    cfg = full_cfg(C)
    P = spec.init_params(cfg, SEED_W)
    plan = spec.make_plan(cfg)
    past, fut = synth_inputs(2, C, H, W, 5, 3, f"full/{gname}/c{C}")
    t = g[f"{key}/t"]
    y = ot.unet_forward(ot.to_torch(P), plan, torch.from_numpy(fut), torch.from_numpy(t), torch.from_numpy(past))
    err = np.abs(y.numpy() - g[f"{key}/out"]).max()
    assert err <= 1e-5, err


@pytest.mark.parametrize("name", ["ethucy", "atc_medium"])
def test_forward_torch_oracle_on_other_shipped_geometries(name):
    """ETH-UCY 8x12 and ATC_medium (16 frames, base 64): the oracle against the reference's own forward (fwd_geoms.npz) --
    the pin behind tests/test_gpu_ref_geometries.py."""
    g = load("fwd_geoms.npz")
    H, W, P_, F, base, att, C = {"ethucy": (8, 12, 5, 3, 32, (False, False, True, False), 3),
                                 "atc_medium": (12, 36, 8, 8, 64, (False, False, True), 4)}[name]
    cfg = spec.UNetConfig(C, C, 1, base, (1, 2, 4), att, 0.1, 4, "Past")
    past, fut = synth_inputs(2, C, H, W, P_, F, f"geom/{name}")
    y = ot.unet_forward(ot.to_torch(spec.init_params(cfg, SEED_W)), spec.make_plan(cfg), torch.from_numpy(fut),
                        torch.from_numpy(g[f"{name}/t"]), torch.from_numpy(past))
    err = np.abs(y.numpy() - g[f"{name}/out"]).max()
    assert err <= 1e-5, err


def test_full_forward_numpy_oracle_atc():
    g = load("fwd.npz")
    cfg = full_cfg(3)
    P = spec.init_params(cfg, SEED_W)
    past, fut = synth_inputs(2, 3, 12, 36, 5, 3, "full/atc/c3")
    y = on.unet_forward(P, spec.make_plan(cfg), fut, g["atc_c3/t"], past)
    err = np.abs(y - g["atc_c3/out"]).max()
    assert err <= TOL / 5, err


def _loop_setup(tag):
    C, H, W, P_, F, B = 3, 12, 36, 5, 3, 2
    cfg = full_cfg(C)
    P = spec.init_params(cfg, SEED_W)
    plan = spec.make_plan(cfg)
    per = C * H * W * F
    past = prng.normal(7, f"past/loop/{tag}", B * C * H * W * P_).reshape(B, C, H, W, P_)
    x_T = prng.normal_per_sample(7, f"xT/{tag}", np.arange(B), per).reshape(B, C, H, W, F)
    Pt = ot.to_torch(P)
    unet = lambda f, t, p: ot.unet_forward(Pt, plan, torch.from_numpy(np.ascontiguousarray(f)),
                                           torch.from_numpy(t), torch.from_numpy(p)).numpy()
    noise = lambda t: loop_noise(tag, B, per, t).reshape(B, C, H, W, F)
    return P, plan, past, x_T, unet, noise


def test_loop_ddpm50():
    g = load("loop.npz")
    P, plan, past, x_T, unet, noise = _loop_setup("ddpm50")
    with torch.inference_mode():
        x, _ = on.generate_ddpm(P, plan, on.schedule(50, 0.5), past, x_T, noise, 50, unet=unet)
    err = np.abs(x - g["ddpm50/x0"]).max()
    assert err <= TOL, err


def test_loop_ddpm20_sparsity():
    g = load("loop.npz")
    P, plan, past, x_T, unet, noise = _loop_setup("ddpm20_sparsity")
    with torch.inference_mode():
        x, _ = on.generate_ddpm(P, plan, on.schedule(20, 0.5), past, x_T, noise, 20, guidance="Sparsity",
                                lam=0.004, unet=unet)
    err = np.abs(x - g["ddpm20_sparsity/x0"]).max()
    assert err <= TOL, err


@pytest.mark.parametrize("guid", ["None", "Sparsity"])
def test_loop_ddim(guid):
    g = load("loop.npz")
    tag = "ddim1000_div100" + ("_sparsity" if guid == "Sparsity" else "")
    P, plan, past, x_T, unet, noise = _loop_setup(tag)
    taus = np.arange(0, 999, 100)
    with torch.inference_mode():
        x = on.generate_ddim(P, plan, on.schedule(1000, 0.5), past, x_T, noise, taus, 1000, 0.001,
                             guidance=guid, lam=0.004, unet=unet)
    scale = max(1.0, np.abs(g[tag + "/x0"]).max())
    err = np.abs(x - g[tag + "/x0"]).max() / scale
    assert err <= TOL, err


def test_loop_ddpm1000_checkpoints():
    """1000-step loop (BASELINE config 2 length, B=2).  The reference differs from
    itself by 5e-5 (8 vs 1 thread) at |x|max ~ 88 here, so the bar is
    relative to |x|max."""
    g = load("loop.npz")
    P, plan, past, x_T, unet, noise = _loop_setup("ddpm1000")
    keep = (999, 900, 500, 0)
    with torch.inference_mode():
        x, kept = on.generate_ddpm(P, plan, on.schedule(1000, 0.5), past, x_T, noise, 1000, keep=keep, unet=unet)
    for t in keep:
        ref = g[f"ddpm1000/x_after_t{t}"]
        err = np.abs(kept[t] - ref).max() / max(1.0, np.abs(ref).max())
        assert err <= 1e-5, (t, err)
    assert np.array_equal(kept[0], x)


def test_metrics_oracle_vs_reference_tables():
    """oracle/metrics_numpy.py against the tables of the reference's MetricsGenerator (tests/golden/metrics.npz)."""
    import warnings
    from crowdmod_ddpm_4d_amd import prng
    from oracle import metrics_numpy as om
    g = load("metrics.npz")
    N, C_, H, W, F, chunk = 8, 3, 12, 36, 3, int(g["chunk"])
    gt = prng.normal(7, "metrics/gt", N * C_ * H * W * F).reshape(N, C_, H, W, F)
    gt[:, 0] = np.maximum(gt[:, 0], 0.0)
    gt[0, 0, :, :, 2] = 0.0
    pred = (gt + 0.3 * prng.normal(7, "metrics/noise", gt.size).reshape(gt.shape)).astype(np.float32)
    for i in range(0, N, chunk):
        gt[i:i + chunk] = gt[i]
    gt = gt.astype(np.float32)
    eps = float(g["eps"])
    np.testing.assert_allclose(om.ranges(gt), g["ranges"], rtol=1e-7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a, mx, ot, mxt = om.psnr_tables(pred, gt, chunk, eps, False)
        am, mxm, otm, mxtm = om.psnr_tables(pred, gt, chunk, eps, True)
    for got, key in ((a, "PSNR"), (mx, "MAX_PSNR"), (ot, "PSNR_OVER_TIME"), (mxt, "MAX_PSNR_OVER_TIME"), (am, "MASK_PSNR"),
                     (mxm, "MAX_MASK_PSNR"), (otm, "MASK_PSNR_OVER_TIME"), (mxtm, "MAX_MASK_PSNR_OVER_TIME")):
        np.testing.assert_allclose(got, g[key], rtol=1e-10, equal_nan=True, err_msg=key)
    re, mre = om.re_density(pred, gt, chunk, eps)
    np.testing.assert_allclose(re, g["RE_DENSITY"], rtol=1e-6)
    np.testing.assert_allclose(mre, g["MIN_RE_DENSITY"], rtol=1e-6)
    np.testing.assert_allclose(om.tv_over_time(pred, gt), g["TV_OVER_TIME"], rtol=1e-6, atol=1e-4)
