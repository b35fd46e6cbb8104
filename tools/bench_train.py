"""BASELINE configs[2]: config/ATC.yml training step (q-sample + UNet fwd + MSE + bwd + Adam), batch 128,
one MI355X, fp32.  Inputs resident in HBM; prints one JSON line.  Not the headline bench (bench.py is)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crowdmod_ddpm_4d_amd import native, prng, spec  # noqa: E402
from crowdmod_ddpm_4d_amd.diffusion import DDPM  # noqa: E402
from crowdmod_ddpm_4d_amd.unet import UNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-update", action="store_true")
    a = ap.parse_args()
    B, Cc, H, W, P, F = a.batch, a.channels, 12, 36, 5, 3
    net = UNet(input_channels=Cc, output_channels=Cc, num_res_blocks=1, base_channels=32,
               base_channels_multiples=(1, 2, 4), apply_attention=(False, False, True), dropout_rate=0.1,
               time_multiple=4, condition="Past")
    net.load_state_dict(spec.init_params(net.cfg, 42))
    net.ensure(H, W, P, F, B)
    net.train_init(lr=5e-5, betas=(0.5, 0.999), weight_decay=0.003)
    sched = DDPM(timesteps=1000, scale=0.5)
    fut = prng.normal(7, "bt/fut", B * Cc * H * W * F).reshape(B, Cc, H, W, F)
    past = prng.normal(7, "bt/past", B * Cc * H * W * P).reshape(B, Cc, H, W, P)
    eps = prng.normal(7, "bt/eps", fut.size).reshape(fut.shape)
    t = (np.arange(B, dtype=np.int64) * 7919) % 1000
    dfut, dpast, deps, dt = (native.DeviceBuffer.from_array(x) for x in (fut, past, eps, t))
    losses = []
    for _ in range(a.warmup):
        losses.append(net.train_step(sched._handle, dfut, dpast, dt, deps, seed=1, apply_update=not a.no_update))
    native.check(native.lib().cm_device_synchronize(0))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses.append(net.train_step(sched._handle, dfut, dpast, dt, deps, seed=1, apply_update=not a.no_update))
    native.check(native.lib().cm_device_synchronize(0))
    dt_s = (time.perf_counter() - t0) / a.steps
    fwd_flops, _ = net.cost(B)
    print(json.dumps({"metric": "train-steps/sec (q-sample + UNet fwd + MSE + bwd + Adam)", "value": 1.0 / dt_s,
                      "unit": "steps/s", "ms_per_step": dt_s * 1e3, "batch": B, "dtype": "f32",
                      "samples_per_s": B / dt_s, "fwd_gflop_per_step": fwd_flops / 1e9,
                      "est_tflops_3x_fwd": 3 * fwd_flops / dt_s / 1e12,
                      "loss_first": losses[0], "loss_last": losses[-1],
                      "config": {"workload": "config/ATC.yml training step (BASELINE configs[2]), B=%d, C=%d" % (B, Cc)}}))


if __name__ == "__main__":
    main()
