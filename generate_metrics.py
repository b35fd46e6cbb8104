#!/usr/bin/env python3
"""Sampling metrics with the reference's command line (generate_metrics.py:73-79 of the reference).

For every test batch: `samples_per_batch / chunk` past windows, each repeated `chunk` times, are sampled in ONE
device loop (cfg MODEL.NSAMPLES = 1280 = 64 pasts x 20 repeats by default), and the per-frame reductions behind
PSNR / masked PSNR / relative density error / total variation run on the device too (cm_frame_metrics); the
tables land in <OUTPUT_DIR>/metrics as CSV + metrics_files.json.  SSIM and ENERGY run on the host (scipy / numpy), as
the reference's do (skimage / torch CPU), and so do the motion-feature histogram metrics MF_MSE / MF_BHATT.

Data: `--data-npy` takes sequences [N, C>=3, ROWS, COLS, T] cut into past/future windows (utils/dataset.py:22-53);
without it synthetic windows ~ N(0,1) with a non-negative density channel are used (no dataset ships here).
"""
import argparse
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from crowdmod_ddpm_4d_amd import config as cfgmod, prng  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description="A script to generate metrics from a trained model (MI355X-native path).")
    ap.add_argument('--chunk-repd-past-seq', type=int, default=None, help='Chunk of repeated past sequences to use when predicting.')
    ap.add_argument('--metric', type=str, default='ALL', help='PSNR|MASK_PSNR|SSIM|MF_MSE|MF_BHATT (= MOTION_FEAT_BHATT)|ENERGY|RE_DENSITY|TV|ALL')
    ap.add_argument('--batches-to-use', type=int, default=1, help='Total of batches to use to compute metrics.')
    ap.add_argument('--config-yml-file', type=str, default='config/ATC.yml')
    ap.add_argument('--configList-yml-file', type=str, default=None)
    ap.add_argument('--model-sample-to-load', type=str, default="000")
    ap.add_argument('--arch', type=str, default='DDPM-UNet')
    ap.add_argument('--data-npy', type=str, default=None, help='test sequences [N,C,ROWS,COLS,T] (.npy)')
    ap.add_argument('--test-windows', type=int, default=None, help='synthetic test windows per batch (default: BATCH_SIZE)')
    ap.add_argument('--timesteps', type=int, default=None, help='override MODEL.DDPM.TIMESTEPS (smoke runs)')
    ap.add_argument('--device', type=int, default=0)
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")
    if args.arch != "DDPM-UNet":
        raise SystemExit(f"{args.arch}: generate_metrics is implemented for DDPM-UNet on this path")
    from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model
    from generate_samples import model_fullname, windows
    cfg = cfgmod.getYamlConfig(args.config_yml_file, args.configList_yml_file)
    if args.timesteps:
        cfg.MODEL.DDPM.TIMESTEPS = int(args.timesteps)
    res = cfgmod.resolve(cfg, args.arch)
    mprops = 3   # generate_metrics.py:60 of the reference
    # generate_metrics.py:63-68: NSAMPLES chains in chunks of 20 repeats, or BATCH_SIZE * chunk when the flag is given
    if args.chunk_repd_past_seq is None:
        samples_per_batch, chunk = res.nsamples, 20
    else:
        samples_per_batch, chunk = res.batch_size * args.chunk_repd_past_seq, args.chunk_repd_past_seq
    out_dir = os.path.join(cfg.DATA_FS.get("OUTPUT_DIR", "output"), "metrics")
    model = DDPM_model(cfg, args.arch, mprops, output_dir=out_dir, device=args.device)
    ckpt = model_fullname(cfg, args.arch, args.model_sample_to_load)
    if os.path.isfile(ckpt):
        logging.info("model full name: %s", ckpt)
        model.load_checkpoint(ckpt)
    else:
        logging.warning("checkpoint %s not found: sampling from randomly initialised weights", ckpt)
    nw = args.test_windows or res.batch_size
    if args.data_npy:
        seq = np.load(args.data_npy).astype(np.float32)
        past, fut = windows(seq, res.past_len, res.future_len, stride=res.past_len + res.future_len, mprops=mprops)
        batches = [(past[i:i + nw], fut[i:i + nw]) for i in range(0, past.shape[0], nw)]
    else:
        batches = []
        for bi in range(args.batches_to_use):
            sp, sf = (nw, mprops, res.rows, res.cols, res.past_len), (nw, mprops, res.rows, res.cols, res.future_len)
            p = prng.normal(11, f"metrics/past/{bi}", int(np.prod(sp))).reshape(sp)
            f = prng.normal(11, f"metrics/future/{bi}", int(np.prod(sf))).reshape(sf)
            p[:, 0], f[:, 0] = np.maximum(p[:, 0], 0), np.maximum(f[:, 0], 0)      # density is non-negative
            batches.append((p, f))
    logging.info("=======>>>> Init metrics compute for %s dataset with %s architecture: %d chains per batch (%d repeats)",
                 cfg.DATASET.get("NAME", "?"), args.arch, samples_per_batch, chunk)
    mg = model.generate_metrics(batches, chunk, args.metric, args.batches_to_use, samples_per_batch, None, out_dir)
    for k, v in mg.data_dict.items():
        if v is not None and len(v):
            logging.info("%-24s %s mean %s", k, np.asarray(v).shape, np.round(np.nanmean(np.asarray(v, dtype=np.float64), axis=0), 3)[:6])
    logging.info("metrics tables in %s", out_dir)
    return mg


if __name__ == '__main__':
    main()
