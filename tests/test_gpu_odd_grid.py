"""A grid none of the kernels is specialised for (8 x 20, full-width UNet): partial tiles of the packed-column and
parity-form weight gradients, the generic (non-Winograd, non-z-split) forward fall-backs, half-resolution planes too small
for a Winograd tile.  No reference fixture exists for this grid, so the checker is the CPU oracle (oracle/unet_torch.py,
itself pinned by the reference fixtures in test_oracle_cpu.py) with torch autograd for the gradients.
Reference lines: models/backbones/unet.py:124-167, models/diffusion/ddpm.py:111-121,142-143."""
import numpy as np
import pytest

from crowdmod_ddpm_4d_amd import prng, spec
from helpers import SEED_W, full_cfg, synth_inputs

pytestmark = pytest.mark.gpu

P_LEN, F_LEN, C_ = 5, 3, 3


def _oracle(params, past, fut, t, eps, masks):
    import torch
    from oracle import unet_torch as ot
    torch.set_num_threads(8)
    cfg = full_cfg(C_)
    plan = spec.make_plan(cfg)
    P = {k: torch.tensor(np.asarray(v), dtype=torch.float32) for k, v in params.items()}
    for k, v in P.items():
        if k != "time_embeddings.time_blocks.0.weight":
            v.requires_grad_(True)
    sched = ot.schedule(1000, scale=0.5)
    tt = torch.tensor(t, dtype=torch.long)
    x0, e = torch.tensor(fut), torch.tensor(eps)
    xt = sched["sqrt_alpha_bar"][tt].view(-1, 1, 1, 1, 1) * x0 + sched["sqrt_one_minus_alpha_bar"][tt].view(-1, 1, 1, 1, 1) * e
    dm = {k: torch.tensor(v) for k, v in masks.items()}
    pred = ot.unet_forward(P, plan, xt, tt, torch.tensor(past), dm)
    loss = torch.nn.functional.mse_loss(pred, e)
    loss.backward()
    with torch.no_grad():
        fwd = ot.unet_forward(P, plan, x0, tt, torch.tensor(past), None)
    return float(loss.detach()), {k: v.grad.numpy() for k, v in P.items() if v.grad is not None}, fwd.numpy()


@pytest.mark.parametrize("H,W,B", [(8, 20, 3), (24, 72, 1)])
def test_odd_grid_forward_and_gradients_vs_torch_oracle(H, W, B):
    """8 x 20: partial tiles and generic fall-backs.  24 x 72 (the doubled grid of BASELINE configs[4]): the training step
    there, including the attention backward at 216 tokens (its S x S matrices live in a global slab, not in LDS)."""
    from crowdmod_ddpm_4d_amd.diffusion import DDPM
    from crowdmod_ddpm_4d_amd.unet import UNet
    cfg = full_cfg(C_)
    params = spec.init_params(cfg, SEED_W)
    past, fut = synth_inputs(B, C_, H, W, P_LEN, F_LEN, "oddgrid")
    eps = prng.normal(11, "oddgrid/eps", fut.size).reshape(fut.shape)
    t = np.array([3, 500, 999])[:B]
    masks = {}
    for blk in spec.make_plan(cfg).res_blocks():
        u = prng.uniform_pm1(11, f"dropodd/{blk.prefix}", B * blk.cout).reshape(B, blk.cout)
        masks[blk.prefix] = ((u * 0.5 + 0.5) >= 0.1).astype(np.float32) / np.float32(0.9)
    ref_loss, ref_grads, ref_fwd = _oracle(params, past, fut, t, eps, masks)

    net = UNet(input_channels=C_, output_channels=C_, num_res_blocks=1, base_channels=32, base_channels_multiples=(1, 2, 4),
               apply_attention=(False, False, True), dropout_rate=0.1, time_multiple=4, condition="Past", max_batch=B)
    net.load_state_dict(params)
    # inference forward (eval mode: no dropout) on the clean future frames
    y = net(fut, t, past)
    assert np.abs(y - ref_fwd).max() <= 1e-4, float(np.abs(y - ref_fwd).max())
    # one training step without the update: loss and every parameter gradient
    net.ensure(H, W, P_LEN, F_LEN, B)
    net.train_init(lr=5e-5, betas=(0.5, 0.999), weight_decay=0.003)
    sampler = DDPM(timesteps=1000, scale=0.5)
    loss = net.train_step(sampler._handle, fut, past, t, eps, drop_masks=masks, apply_update=False)
    assert abs(loss - ref_loss) <= 1e-5 * max(1.0, ref_loss), (loss, ref_loss)
    worst = ("", 0.0)
    for name, gref in ref_grads.items():
        got = net.grad(name).astype(np.float64).reshape(gref.shape)
        err = float(np.abs(got - gref).max())
        scale = float(np.abs(gref).max())
        rel = err / (scale + 1e-12)
        if rel > worst[1]:
            worst = (name, rel)
        assert err <= 3e-4 * scale + 2e-8, (name, err, scale)
    assert len(ref_grads) >= 160, len(ref_grads)
    print("worst relative gradient error", worst)
