"""Sampling-step time on another reference grid (not the headline bench): python tools/bench_grid.py config/HERMES-CR-120.yml [B] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crowdmod_ddpm_4d_amd import config as cfgmod, native, prng  # noqa: E402
from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model  # noqa: E402
from crowdmod_ddpm_4d_amd.diffusion import DDPM  # noqa: E402

path = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rows = cols = None
if len(sys.argv) > 5:
    rows, cols = int(sys.argv[4]), int(sys.argv[5])
cfg = cfgmod.getYamlConfig(path, None)
if rows:
    cfg.MACROPROPS.ROWS, cfg.MACROPROPS.COLS = rows, cols
res = cfgmod.resolve(cfg, "DDPM-UNet")
m = DDPM_model(cfg, "DDPM-UNet", 3)
m.denoiser.max_batch = B
s = DDPM(timesteps=res.timesteps, scale=res.scale)
past = prng.normal(7, "grid/past", B * 3 * res.rows * res.cols * res.past_len).reshape(B, 3, res.rows, res.cols, res.past_len)
m._generate_ddpm(past, s, B, first_steps=5)
native.check(native.lib().cm_device_synchronize(0))
t0 = time.perf_counter()
m._generate_ddpm(past, s, B, first_steps=steps)
native.check(native.lib().cm_device_synchronize(0))
dt = (time.perf_counter() - t0) / steps
fl, _ = m.denoiser.cost(B)
print(f"{path} grid {res.rows}x{res.cols} B={B}: {dt * 1e3:.3f} ms/step, {1 / dt:.1f} steps/s, {fl / dt / 1e12:.1f} TFLOP/s (host-staged call, {steps} steps)")
