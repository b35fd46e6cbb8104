"""Shared helpers for the parity tests: synthetic inputs identical to the ones
tests/golden/make_golden.py fed to the reference."""
import os

import numpy as np

from crowdmod_ddpm_4d_amd import prng, spec

SEED_W, SEED_X = 42, 7
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NARROW = dict(H=4, W=8, P=5, F=3, B=2)
FULL_GRIDS = {"atc": (12, 36), "cr120": (28, 24), "atc2x": (24, 72)}


def narrow_cfg(C):
    return spec.UNetConfig(C, C, 1, 8, (1, 2, 4), (False, False, True, False), 0.1, 4, "Past")


def full_cfg(C):
    return spec.UNetConfig(C, C, 1, 32, (1, 2, 4), (False, False, True, False), 0.1, 4, "Past")


def synth_inputs(B, C, H, W, P, F, tag):
    past = prng.normal(SEED_X, f"past/{tag}", B * C * H * W * P).reshape(B, C, H, W, P)
    fut = prng.normal(SEED_X, f"future/{tag}", B * C * H * W * F).reshape(B, C, H, W, F)
    return past, fut


def loop_noise(tag, B, per, t):
    return prng.normal_per_sample(SEED_X, f"z/{tag}", np.arange(B), per, step=t)


def load(name):
    return np.load(os.path.join(GOLDEN, name))
