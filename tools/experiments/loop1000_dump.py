"""Run the full T = 1000 DDPM reverse loop (ATC geometry, B = 2, injected x_T and per-step noise) and one forward on the library
named by CM_LIB_PATH; save the results so that two builds can be compared (tools/experiments/loop1000_cmp.py).
  python tools/experiments/loop1000_dump.py <out.npz> [T]"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from crowdmod_ddpm_4d_amd import prng, spec  # noqa: E402
from crowdmod_ddpm_4d_amd.config import AttrDict  # noqa: E402
from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model  # noqa: E402
from crowdmod_ddpm_4d_amd.diffusion import DDPM  # noqa: E402
from helpers import SEED_W, synth_inputs  # noqa: E402

out = sys.argv[1]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
B, C, H, W, P, F = 2, 4, 12, 36, 5, 3
cfg = AttrDict({
    "MACROPROPS": {"ROWS": H, "COLS": W}, "DATASET": {"PAST_LEN": P, "FUTURE_LEN": F, "BATCH_SIZE": B},
    "MODEL": {"NSAMPLES": B, "NSAMPLES4PLOTS": B, "DDPM": {
        "SAMPLER": "DDPM", "TIMESTEPS": T, "SCALE": 0.5, "SIGMA": 0.001, "DDIM_DIVIDER": 2, "GUIDANCE": "None", "LAMBDA_GUIDANCE": 0.0,
        "UNET": {"CONDITION": "Past", "NUM_RES_BLOCKS": 1, "BASE_CH": 32, "BASE_CH_MULT": [1, 2, 4],
                 "APPLY_ATTENTION": [False, False, True, False], "DROPOUT_RATE": 0.1, "TIME_EMB_MULT": 4}}}})
ucfg = spec.UNetConfig(C, C, 1, 32, (1, 2, 4), (False, False, True, False), 0.1, 4, "Past")
m = DDPM_model(cfg, "DDPM-UNet", C)
m.denoiser.load_state_dict(spec.init_params(ucfg, SEED_W))
past, fut = synth_inputs(B, C, H, W, P, F, "loop1000")
t = np.array([3, 777])
y = m.denoiser(fut, t, past)
per = C * H * W * F
x_T = prng.normal_per_sample(7, "loop1000/xT", np.arange(B), per).reshape(B, C, H, W, F)
noise = np.stack([prng.normal_per_sample(7, "loop1000/z", np.arange(B), per, step=s).reshape(B, C, H, W, F) for s in range(T - 1, 0, -1)])
x, _ = m._generate_ddpm(past, DDPM(timesteps=T, scale=0.5), B, x_T=x_T, noise=noise)
np.savez(out, fwd=y, x0=x)
print("saved", out, "fwd absmax %.4f x0 absmax %.4f" % (float(np.abs(y).max()), float(np.abs(x).max())))
