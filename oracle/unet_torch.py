"""ORACLE (test infrastructure, not product code) -- torch.nn.functional CPU
restatement of the reference's DDPM-UNet hot path.

Same algorithm as oracle/unet_numpy.py, written against ATen's CPU operators
(conv3d / group_norm / silu / interpolate / scaled_dot_product_attention) --
i.e. the same CPU kernels the reference's nn.Modules dispatch to
(/root/reference/models/backbones/layers.py, unet.py, embeddings.py).  It is
this repo's own code: no reference file is imported.  Uses:

  * the fast checker at sizes where the NumPy oracle is too slow;
  * bench.py's `cpu_baseline` leg ("port", multi-threaded, cores stated).

Pinned against the same golden vectors as the NumPy oracle
(tests/test_oracle_golden.py).  Never imported by the product path.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

GN_GROUPS = 8
HEADS = 4


def to_torch(P, dtype=torch.float32):
    return {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in P.items()}


def _res_block(x, temb, P, pre, drop_mask=None):
    """layers.py:55-78."""
    h = F.silu(F.group_norm(x, GN_GROUPS, P[pre + ".normalize_1.weight"], P[pre + ".normalize_1.bias"]))
    h = F.conv3d(h, P[pre + ".conv_1.weight"], P[pre + ".conv_1.bias"], padding=1)
    h = h + F.linear(F.silu(temb), P[pre + ".dense_1.weight"], P[pre + ".dense_1.bias"])[:, :, None, None, None]
    h = F.silu(F.group_norm(h, GN_GROUPS, P[pre + ".normalize_2.weight"], P[pre + ".normalize_2.bias"]))
    if drop_mask is not None:
        h = h * drop_mask[:, :, None, None, None]
    h = F.conv3d(h, P[pre + ".conv_2.weight"], P[pre + ".conv_2.bias"], padding=1)
    if (pre + ".match_input.weight") in P:
        h = h + F.conv3d(x, P[pre + ".match_input.weight"], P[pre + ".match_input.bias"])
    else:
        h = h + x
    if (pre + ".attention.group_norm.weight") in P:
        h = _attention(h, P, pre + ".attention")
    return h


def _attention(x, P, pre):
    """layers.py:12-18 (MHA restated via packed projection + SDPA)."""
    B, C, H, W, L = x.shape
    d = C // HEADS
    h = F.group_norm(x, GN_GROUPS, P[pre + ".group_norm.weight"], P[pre + ".group_norm.bias"])
    h = h.reshape(B, C, H * W * L).swapaxes(1, 2)
    qkv = F.linear(h, P[pre + ".mhsa.in_proj_weight"], P[pre + ".mhsa.in_proj_bias"])
    q, k, v = qkv.split(C, dim=-1)
    sp = lambda t: t.reshape(B, -1, HEADS, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), scale=1.0 / math.sqrt(d))
    o = o.transpose(1, 2).reshape(B, -1, C)
    o = F.linear(o, P[pre + ".mhsa.out_proj.weight"], P[pre + ".mhsa.out_proj.bias"])
    return x + o.swapaxes(2, 1).reshape(B, C, H, W, L)


def time_embedding(t, P):
    """embeddings.py:33-34."""
    e = P["time_embeddings.time_blocks.0.weight"][t]
    e = F.silu(F.linear(e, P["time_embeddings.time_blocks.1.weight"], P["time_embeddings.time_blocks.1.bias"]))
    return F.linear(e, P["time_embeddings.time_blocks.3.weight"], P["time_embeddings.time_blocks.3.bias"])


def unet_forward(P, plan, future, t, past, drop_masks=None):
    """unet.py:124-167.  P: dict of torch tensors (see to_torch)."""
    temb = time_embedding(t, P)
    past_len = past.shape[4]
    x = torch.cat([past, future], dim=4)
    h = F.conv3d(x, P["first.weight"], P["first.bias"], padding=1)
    outs = [h]

    def run(blk, h):
        if blk.kind == "res":
            dm = None if drop_masks is None else drop_masks.get(blk.prefix)
            return _res_block(h, temb, P, blk.prefix, dm)
        if blk.kind == "down":
            return F.conv3d(h, P[blk.prefix + ".downsample.weight"], P[blk.prefix + ".downsample.bias"],
                            stride=2, padding=1)
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        return F.conv3d(h, P[blk.prefix + ".upsample.1.weight"], P[blk.prefix + ".upsample.1.bias"], padding=1)

    for blk in plan.encoder:
        h = run(blk, h)
        outs.append(h)
    for blk in plan.bottleneck:
        h = run(blk, h)
    for blk in plan.decoder:
        if blk.kind == "res":
            h = torch.cat([h, outs.pop()], dim=1)
        h = run(blk, h)
    h = F.silu(F.group_norm(h, GN_GROUPS, P["final.0.weight"], P["final.0.bias"]))
    h = F.conv3d(h, P["final.2.weight"], P["final.2.bias"], padding=1)
    return h[:, :, :, :, past_len:]


def schedule(timesteps=1000, scale=1.0, beta_start=1e-4, beta_end=2e-2):
    """forward.py:10-27."""
    beta = torch.linspace(scale * beta_start, scale * beta_end, timesteps, dtype=torch.float32)
    alpha = 1 - beta
    abar = torch.cumprod(alpha, dim=0)
    return {"beta": beta, "alpha": alpha, "alpha_bar": abar, "sqrt_alpha_bar": torch.sqrt(abar),
            "one_by_sqrt_alpha": 1.0 / torch.sqrt(alpha), "sqrt_one_minus_alpha_bar": torch.sqrt(1 - abar)}


def ddpm_step(sched, eps_hat, x, t, z):
    """ddpm.py:25-38 with z injected."""
    beta = sched["beta"][t]
    return sched["one_by_sqrt_alpha"][t] * (x - (beta / sched["sqrt_one_minus_alpha_bar"][t]) * eps_hat) \
        + torch.sqrt(beta) * z


@torch.inference_mode()
def generate_ddpm(P, plan, sched, past, x_T, noise_fn, timesteps, t_list=None):
    """ddpm.py:206-236 with injected x_T / z_t.  `t_list` restricts the loop to a
    given descending list of t (used to time a bounded sample of steps)."""
    x = x_T.clone()
    ts = list(reversed(range(timesteps))) if t_list is None else list(t_list)
    for t in ts:
        tt = torch.full((x.shape[0],), t, dtype=torch.long)
        eps_hat = unet_forward(P, plan, x, tt, past)
        z = noise_fn(t) if t > 0 else torch.zeros_like(x)
        x = ddpm_step(sched, eps_hat, x, t, z)
    return x
