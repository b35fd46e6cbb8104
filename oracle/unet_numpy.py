"""ORACLE (test infrastructure, not product code) -- NumPy restatement of the
reference's DDPM-UNet hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this module; the product path (crowdmod-ddpm-4d_amd/) never does
and fails loudly when its HIP library is missing.

Parity pin: this restatement is checked against golden vectors captured from
the reference's own modules imported in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz; see
tests/test_oracle_golden.py).  The reference has no tests or known-answer
vectors of its own.

All tensors use the reference layout [B, C, H, W, L] (L = frames innermost,
/root/reference/utils/dataset.py:48-51).  `dtype` selects the arithmetic
precision: float32 restates the reference, float64 gives a ground truth to
judge which of two fp32 implementations is closer.

Every function cites the reference lines it restates.
"""
from __future__ import annotations

import numpy as np

GN_GROUPS = 8
GN_EPS = 1e-5
HEADS = 4


# --------------------------------------------------------------------------- #
# elementwise / small ops
# --------------------------------------------------------------------------- #
def silu(x):
    """nn.SiLU: x * sigmoid(x)  (layers.py:27, embeddings.py:28)."""
    return x / (1.0 + np.exp(-x))


def linear(x, w, b):
    """nn.Linear: x @ w.T + b  (embeddings.py:26,30; layers.py:35)."""
    return x @ w.T + b


def group_norm(x, gamma, beta, groups=GN_GROUPS, eps=GN_EPS):
    """nn.GroupNorm(8, C): biased variance over (C/G, H, W, L) per (sample, group)
    (layers.py:9,30,41; unet.py:119)."""
    B, C = x.shape[:2]
    xs = x.reshape(B, groups, -1)
    mean = xs.mean(axis=2, keepdims=True)
    var = ((xs - mean) ** 2).mean(axis=2, keepdims=True)
    y = ((xs - mean) / np.sqrt(var + x.dtype.type(eps))).reshape(x.shape)
    shape = (1, C) + (1,) * (x.ndim - 2)
    return y * gamma.reshape(shape) + beta.reshape(shape)


def upsample_nearest2(x):
    """nn.Upsample(scale_factor=2, mode='nearest') on the three trailing dims
    (layers.py:93): out[i] = in[i // 2]."""
    return x.repeat(2, axis=2).repeat(2, axis=3).repeat(2, axis=4)


def conv3d(x, w, b, stride=1, pad=1):
    """nn.Conv3d cross-correlation with zero padding (layers.py:32,43,46,84,94;
    unet.py:32,121).  x [B,Ci,H,W,L], w [Co,Ci,kH,kW,kL] -> [B,Co,H',W',L']."""
    B, Ci = x.shape[:2]
    Co, _, kh, kw, kl = w.shape
    if pad:
        x = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad), (pad, pad)))
    win = np.lib.stride_tricks.sliding_window_view(x, (kh, kw, kl), axis=(2, 3, 4))
    win = win[:, :, ::stride, ::stride, ::stride]          # [B,Ci,H',W',L',kh,kw,kl]
    Ho, Wo, Lo = win.shape[2:5]
    out = np.empty((B, Co, Ho, Wo, Lo), dtype=x.dtype)
    wmat = w.reshape(Co, -1)                                # [Co, Ci*k^3]
    for bi in range(B):                                     # bound im2col memory
        cols = win[bi].transpose(1, 2, 3, 0, 4, 5, 6).reshape(Ho * Wo * Lo, -1)
        out[bi] = (cols @ wmat.T + b).T.reshape(Co, Ho, Wo, Lo)
    return out


def mha_self(x, in_w, in_b, out_w, out_b, heads=HEADS):
    """nn.MultiheadAttention(E, 4, batch_first=True)(x, x, x) without dropout
    (layers.py:10,16): packed in_proj rows ordered q,k,v; heads split
    contiguously over E; softmax(q k^T / sqrt(d)) v; out_proj.  x [B,S,E]."""
    B, S, E = x.shape
    d = E // heads
    qkv = x @ in_w.T + in_b
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]

    def split(t):
        return t.reshape(B, S, heads, d).transpose(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    s = (q @ k.transpose(0, 1, 3, 2)) * x.dtype.type(1.0 / np.sqrt(d))
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(axis=-1, keepdims=True)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(B, S, E)
    return o @ out_w.T + out_b


# --------------------------------------------------------------------------- #
# blocks
# --------------------------------------------------------------------------- #
def attention_block(x, P, pre):
    """AttentionBlock.forward (layers.py:12-18): x + MHA(GN(x)) over H*W*L tokens."""
    B, C, H, W, L = x.shape
    h = group_norm(x, P[pre + ".group_norm.weight"], P[pre + ".group_norm.bias"])
    h = h.reshape(B, C, H * W * L).swapaxes(1, 2)
    h = mha_self(h, P[pre + ".mhsa.in_proj_weight"], P[pre + ".mhsa.in_proj_bias"],
                 P[pre + ".mhsa.out_proj.weight"], P[pre + ".mhsa.out_proj.bias"])
    h = h.swapaxes(2, 1).reshape(B, C, H, W, L)
    return x + h


def resnet_block(x, temb, P, pre, drop_mask=None, trace=None):
    """ResnetBlock.forward (layers.py:55-78).  `drop_mask` [B,Cout] holds the
    Dropout3d keep-mask already scaled by 1/(1-p) (training); None = eval."""
    h = silu(group_norm(x, P[pre + ".normalize_1.weight"], P[pre + ".normalize_1.bias"]))
    h = conv3d(h, P[pre + ".conv_1.weight"], P[pre + ".conv_1.bias"])
    h = h + linear(silu(temb), P[pre + ".dense_1.weight"], P[pre + ".dense_1.bias"])[:, :, None, None, None]
    h = silu(group_norm(h, P[pre + ".normalize_2.weight"], P[pre + ".normalize_2.bias"]))
    if drop_mask is not None:
        h = h * drop_mask[:, :, None, None, None]
    h = conv3d(h, P[pre + ".conv_2.weight"], P[pre + ".conv_2.bias"])
    if (pre + ".match_input.weight") in P:
        h = h + conv3d(x, P[pre + ".match_input.weight"], P[pre + ".match_input.bias"], pad=0)
    else:
        h = h + x
    if (pre + ".attention.group_norm.weight") in P:
        h = attention_block(h, P, pre + ".attention")
    return h


def time_embedding(t, P):
    """SinusoidalPositionEmbeddings.forward (embeddings.py:33-34): table[t] ->
    Linear -> SiLU -> Linear."""
    e = P["time_embeddings.time_blocks.0.weight"][np.asarray(t, dtype=np.int64)]
    e = linear(e, P["time_embeddings.time_blocks.1.weight"], P["time_embeddings.time_blocks.1.bias"])
    e = silu(e)
    return linear(e, P["time_embeddings.time_blocks.3.weight"], P["time_embeddings.time_blocks.3.bias"])


def unet_forward(P, plan, future, t, past, dtype=np.float32, drop_masks=None, trace=None):
    """UNet.forward (unet.py:124-167).  `plan` is spec.make_plan(cfg); `P` the
    state_dict as numpy arrays.  Returns eps_hat [B,C,H,W,F]."""
    P = {k: np.asarray(v, dtype=dtype) for k, v in P.items()}
    future = np.asarray(future, dtype=dtype)
    past = np.asarray(past, dtype=dtype)
    temb = time_embedding(t, P)
    past_len = past.shape[4]
    x = np.concatenate([past, future], axis=4)
    h = conv3d(x, P["first.weight"], P["first.bias"])
    outs = [h]

    def run(block, h):
        if block.kind == "res":
            dm = None if drop_masks is None else drop_masks.get(block.prefix)
            return resnet_block(h, temb, P, block.prefix, dm)
        if block.kind == "down":
            return conv3d(h, P[block.prefix + ".downsample.weight"], P[block.prefix + ".downsample.bias"], stride=2)
        return conv3d(upsample_nearest2(h), P[block.prefix + ".upsample.1.weight"], P[block.prefix + ".upsample.1.bias"])

    for blk in plan.encoder:
        h = run(blk, h)
        outs.append(h)
        if trace is not None:
            trace[blk.prefix] = h
    for blk in plan.bottleneck:
        h = run(blk, h)
        if trace is not None:
            trace[blk.prefix] = h
    for blk in plan.decoder:
        if blk.kind == "res":
            h = np.concatenate([h, outs.pop()], axis=1)
        h = run(blk, h)
        if trace is not None:
            trace[blk.prefix] = h
    h = silu(group_norm(h, P["final.0.weight"], P["final.0.bias"]))
    h = conv3d(h, P["final.2.weight"], P["final.2.bias"])
    return h[:, :, :, :, past_len:]


# --------------------------------------------------------------------------- #
# diffusion schedule, forward noising, reverse steps
# --------------------------------------------------------------------------- #
def linspace_f32(start, end, steps):
    """torch.linspace(start, end, steps, dtype=float32) on CPU, bit for bit
    (checked against the golden tables): the step is rounded to fp32, the
    sequence is filled symmetrically from both ends, and each element is a
    fused multiply-add (one rounding) -- reproduced here by forming the exact
    fp32 x int product in float64 and rounding once.  Feeds forward.py:15-20."""
    start32, end32 = np.float32(start), np.float32(end)
    step = np.float64(np.float32((float(end32) - float(start32)) / (steps - 1)))
    i = np.arange(steps, dtype=np.int64)
    lo = np.float64(start32) + step * i
    hi = np.float64(end32) - step * (steps - 1 - i)
    return np.where(i < steps // 2, lo, hi).astype(np.float32)


def schedule(timesteps=1000, scale=1.0, beta_start=1e-4, beta_end=2e-2):
    """ForwardSampler.__init__ (forward.py:10-27): six [T] fp32 tables."""
    beta = linspace_f32(scale * beta_start, scale * beta_end, timesteps)
    alpha = (np.float32(1) - beta).astype(np.float32)
    # torch.cumprod on CPU accumulates fp32 inputs in double (ATen acc_type)
    # and rounds each prefix product to fp32 -- bit-exact vs the golden tables.
    alpha_bar = np.cumprod(alpha.astype(np.float64)).astype(np.float32)
    return {
        "beta": beta,
        "alpha": alpha,
        "alpha_bar": alpha_bar,
        "sqrt_alpha_bar": np.sqrt(alpha_bar).astype(np.float32),
        "one_by_sqrt_alpha": (np.float32(1) / np.sqrt(alpha)).astype(np.float32),
        "sqrt_one_minus_alpha_bar": np.sqrt(np.float32(1) - alpha_bar).astype(np.float32),
    }


def q_sample(sched, x0, t, eps):
    """ForwardSampler.forward (forward.py:29-36) with the noise injected:
    x_t = sqrt(abar[t]) * x0 + sqrt(1-abar[t]) * eps, per-sample t."""
    t = np.asarray(t, dtype=np.int64)
    a = sched["sqrt_alpha_bar"][t].reshape(-1, 1, 1, 1, 1).astype(x0.dtype)
    s = sched["sqrt_one_minus_alpha_bar"][t].reshape(-1, 1, 1, 1, 1).astype(x0.dtype)
    return a * x0 + s * eps


def ddpm_step(sched, eps_hat, x, t, z):
    """DDPM.step (ddpm.py:25-38) with z injected (z must be 0 when t == 0).
    Returns (x_prev, sqrt(beta_t), 1 - beta_t)."""
    dt = x.dtype.type
    beta = dt(sched["beta"][t])
    c1 = dt(sched["one_by_sqrt_alpha"][t])
    s1m = dt(sched["sqrt_one_minus_alpha_bar"][t])
    xd = c1 * (x - (beta / s1m) * eps_hat) + np.sqrt(beta) * z
    return xd, np.sqrt(beta), dt(1) - beta


def sparsity_gradient(x):
    """sparsityGradient (guidance.py:4-8): sign(x) on channel 0, zero elsewhere."""
    g = np.zeros_like(x)
    g[:, 0] = np.sign(x[:, 0])
    return g


def generate_ddpm(P, plan, sched, past, x_T, noise_fn, timesteps, dtype=np.float32,
                  guidance="None", lam=0.0, keep=None, unet=None):
    """DDPM_model._generate_ddpm (ddpm.py:206-236) with x_T and z_t injected.
    `noise_fn(t)` returns z_t for t > 0.  `keep` = iterable of t whose post-step
    state is recorded.  `unet` overrides the denoiser callable (future,t,past)."""
    unet = unet or (lambda f, t, p: unet_forward(P, plan, f, t, p, dtype=dtype))
    x = np.asarray(x_T, dtype=dtype)
    past = np.asarray(past, dtype=dtype)
    kept = {}
    for t in reversed(range(timesteps)):
        tt = np.full((x.shape[0],), t, dtype=np.int64)
        eps_hat = unet(x, tt, past)
        z = np.asarray(noise_fn(t), dtype=dtype) if t > 0 else np.zeros_like(x)
        x, sigma, _ = ddpm_step(sched, eps_hat, x, t, z)
        if guidance == "Sparsity":                           # ddpm.py:223-226
            x = x - dtype(lam) * sigma * sparsity_gradient(x)
        if keep is not None and t in keep:
            kept[t] = x.copy()
    return x, kept


def generate_ddim(P, plan, sched, past, x_T, noise_fn, taus, timesteps, sigma_t,
                  dtype=np.float32, guidance="None", lam=0.0, unet=None):
    """DDPM_model._generate_ddim (ddpm.py:238-282): DDIM Eq. 12 over reversed(taus),
    carrying the schedule values of the previously visited step; noise is drawn on
    every step including the last (ddpm.py:264)."""
    unet = unet or (lambda f, t, p: unet_forward(P, plan, f, t, p, dtype=dtype))
    x = np.asarray(x_T, dtype=dtype)
    past = np.asarray(past, dtype=dtype)
    last = timesteps - 1
    beta_t = dtype(sched["beta"][last])
    sab_t = dtype(sched["sqrt_alpha_bar"][last])
    s1m_t = dtype(sched["sqrt_one_minus_alpha_bar"][last])
    sig = dtype(sigma_t)
    for t in reversed(list(taus)):
        tt = np.full((x.shape[0],), t, dtype=np.int64)
        eps_hat = unet(x, tt, past)
        beta_p = dtype(sched["beta"][t])
        sab_p = dtype(sched["sqrt_alpha_bar"][t])
        s1m_p = dtype(sched["sqrt_one_minus_alpha_bar"][t])
        x0 = (x - s1m_t * eps_hat) / sab_t
        dirx = np.sqrt(dtype(1) - sab_p ** 2 - sig ** 2) * eps_hat
        x = sab_p * x0 + dirx + sig * np.asarray(noise_fn(t), dtype=dtype)
        if guidance == "Sparsity":                           # ddpm.py:267-271
            x = x - dtype(lam) * np.sqrt(beta_t) * sparsity_gradient(x)
        beta_t, sab_t, s1m_t = beta_p, sab_p, s1m_p
    return x


def mse_loss(a, b):
    """F.mse_loss default reduction='mean' (ddpm.py:120)."""
    return np.mean((a - b) ** 2, dtype=a.dtype)
