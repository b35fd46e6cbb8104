#!/usr/bin/env python3
"""Training entry point with the reference's command line (train.py:73-82).

Round-1 status: the forward-noising kernel (q-sample) and the denoiser forward are
native; the backward pass, Dropout3d and the fused Adam step are NOT built yet
(SURVEY.md section 8 row a13 -- scheduled after the sampling path).  This script parses
the same flags and configuration, then stops with an explicit message instead of
silently falling back to another implementation.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from crowdmod_ddpm_4d_amd import config as cfgmod  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="Train a crowd-macroprops model (MI355X-native path).")
    ap.add_argument('--config-yml-file', type=str, default='config/ATC.yml')
    ap.add_argument('--configList-yml-file', type=str, default=None)
    ap.add_argument('--arch', type=str, default='DDPM-UNet')
    ap.add_argument('--baseline-ckpt', type=str, default=None)
    args = ap.parse_args()
    cfg = cfgmod.getYamlConfig(args.config_yml_file, args.configList_yml_file)
    res = cfgmod.resolve(cfg, args.arch)
    raise SystemExit(
        f"train.py: parsed {args.config_yml_file} ({res.rows}x{res.cols}, base {res.base_ch}, T={res.timesteps}); "
        "the native training step (UNet backward + Adam) is not implemented yet -- there is deliberately no "
        "fallback to another framework.")


if __name__ == '__main__':
    main()
