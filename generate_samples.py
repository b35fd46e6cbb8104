#!/usr/bin/env python3
"""Sample future crowd-macroproperty frames with the MI355X-native DDPM-UNet path.

Keeps the command line of the reference's generate_samples.py (same flag names and
defaults where they make sense here); the plotting flags are accepted and ignored --
this entry point writes the sampled tensor to `<OUTPUT_DIR>/predictions.npz`
(predictions [N,C,H,W,F], past_idx, pasts, futures) for the caller's own plotting.

Data: `--data-npy` takes an array [N, C>=mprops, ROWS, COLS, T] (the reference's
in-memory format, utils/dataset.py:119) that is cut into past/future windows; without
it, synthetic pasts ~ N(0,1) are used (there is no dataset in this repository).
"""
import argparse
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from crowdmod_ddpm_4d_amd import config as cfgmod, prng  # noqa: E402
from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model  # noqa: E402


def model_fullname(cfg, arch, epoch_tag):
    """utils/utils.py:149-167: SAVE_DIR + NAME.format(arch, EPOCHS, PAST_LEN, FUTURE_LEN, tag, 'NA')."""
    res = cfgmod.resolve(cfg, arch)
    epochs = res.train.EPOCHS if res.train is not None and "EPOCHS" in res.train else 0
    name = cfg.MODEL.NAME.format(arch, epochs, res.past_len, res.future_len, epoch_tag, "NA")
    return os.path.join(cfg.DATA_FS.SAVE_DIR, name)


def windows(seq, past_len, future_len, stride, mprops):
    """MacropropsDataset (utils/dataset.py:22-53): sliding windows over the last axis."""
    n, _, _, _, total = seq.shape
    w = past_len + future_len
    out_p, out_f = [], []
    for i in range(n):
        for t in range(0, total - w + 1, stride):
            win = seq[i, :mprops, :, :, t:t + w]
            out_p.append(win[..., :past_len])
            out_f.append(win[..., past_len:])
    return np.stack(out_p), np.stack(out_f)


def main():
    ap = argparse.ArgumentParser(description="Sample crowd macroprops from a trained DDPM-UNet (MI355X-native path).")
    ap.add_argument('--plot-mprop', type=str, default="Density&Vel")
    ap.add_argument('--plot-past', type=str, default='Last2')
    ap.add_argument('--vel-scale', type=float, default=0.5)
    ap.add_argument('--headwidth', type=int, default=5)
    ap.add_argument('--vel-unc-scale', type=int, default=1)
    ap.add_argument('--plot-type', type=str, default='Static')
    ap.add_argument('--same-past-seq', type=bool, default=False)
    ap.add_argument('--config-yml-file', type=str, default='config/ATC.yml')
    ap.add_argument('--configList-yml-file', type=str, default=None)
    ap.add_argument('--model-sample-to-load', type=str, default="000")
    ap.add_argument('--arch', type=str, default='DDPM-UNet')
    ap.add_argument('--from-fixed-past', type=bool, default=False)
    ap.add_argument('--data-npy', type=str, default=None, help='test sequences [N,C,ROWS,COLS,T] (.npy)')
    ap.add_argument('--device', type=int, default=0)
    args = ap.parse_args()
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")

    if args.arch not in ("DDPM-UNet", "FM-UNet"):
        raise SystemExit(f"{args.arch}: only the UNet-backbone generators (DDPM-UNet, FM-UNet) are implemented on this path")
    cfg = cfgmod.getYamlConfig(args.config_yml_file, args.configList_yml_file)
    res = cfgmod.resolve(cfg, args.arch)
    mprops = 3  # generate_samples.py:76 of the reference
    if args.arch == "FM-UNet":
        from crowdmod_ddpm_4d_amd.flow_matching import FM_model as Model
    else:
        Model = DDPM_model
    model = Model(cfg, args.arch, mprops, output_dir=cfg.DATA_FS.get("OUTPUT_DIR", "output"),
                  from_fixed_past=args.from_fixed_past, device=args.device)
    ckpt = model.checkpoint_path(args.model_sample_to_load) if args.arch == "FM-UNet" else \
        model_fullname(cfg, args.arch, args.model_sample_to_load)
    if os.path.isfile(ckpt):
        logging.info("model full name: %s", ckpt)
        model.load_checkpoint(ckpt)
    else:
        logging.warning("checkpoint %s not found: sampling from randomly initialised weights", ckpt)
    if args.data_npy:
        seq = np.load(args.data_npy).astype(np.float32)
        past, fut = windows(seq, res.past_len, res.future_len, stride=res.past_len + res.future_len, mprops=mprops)
        past, fut = past[:res.batch_size], fut[:res.batch_size]
    else:
        n = res.batch_size
        shape_p = (n, mprops, res.rows, res.cols, res.past_len)
        shape_f = (n, mprops, res.rows, res.cols, res.future_len)
        past = prng.normal(7, "cli/past", int(np.prod(shape_p))).reshape(shape_p)
        fut = prng.normal(7, "cli/future", int(np.prod(shape_f))).reshape(shape_f)
    pred, idx, pasts, futures = model.sampling([(past, fut)], args.plot_type, None, args.plot_mprop, args.plot_past,
                                               args.same_past_seq, None)
    os.makedirs(model.output_dir, exist_ok=True)
    out = os.path.join(model.output_dir, "predictions.npz")
    np.savez_compressed(out, predictions=pred, past_idx=idx, pasts=pasts, futures=futures)
    logging.info("sampled %s -> %s", pred.shape, out)


if __name__ == '__main__':
    main()
