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


def join_all(procs, timeout):
    """Join child processes; a child that is still alive after `timeout` seconds (hung rendezvous, hung rank) is
    terminated -- then killed -- before the caller asserts, so it cannot keep the GPU or the TCP port for the rest of
    the pytest session.  Returns the exit codes (None never: a killed child reports its signal)."""
    import time
    deadline = time.monotonic() + timeout
    try:
        for p in procs:
            p.join(timeout=max(0.0, deadline - time.monotonic()))
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
            if p.is_alive():
                p.kill()
                p.join()
    return [p.exitcode for p in procs]
