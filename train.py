#!/usr/bin/env python3
"""Training entry point with the reference's command line (train.py:75-82 of the reference).

Every batch is ONE native call on the MI355X (cm_train_step): q-sample with device-drawn noise,
train-mode UNet forward (Dropout3d), MSE, backward, Adam with coupled L2, weight re-pack.  The host
keeps what the reference's loop keeps on the host: epoch bookkeeping, ReduceLROnPlateau, NaN early
stop, checkpoints in the reference's {"opt", "model"} torch-zip format and file naming.

Data: `--data-npy` takes sequences [N, C>=3, ROWS, COLS, T] (the reference's in-memory format,
utils/dataset.py:119) cut into sliding past/future windows; without it a synthetic set ~ N(0,1) of
`--synthetic-samples` windows is used (there is no dataset in this repository, and no W&B: the
per-epoch record goes to the log and to <SAVE_DIR>/train_log.jsonl).
With WORLD_SIZE > 1 (python -m torch.distributed.run) the batch stream is sharded over ranks and the
flat gradient buffer is averaged with one RCCL all-reduce per step.
"""
import argparse
import json
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from crowdmod_ddpm_4d_amd import config as cfgmod, prng  # noqa: E402


def make_loader(past, future, batch_size, seed, rank=0, world=1):
    """Shuffled mini-batches, drop_last like a fixed-geometry device step wants; rank r of `world`
    takes every world-th batch of the common permutation."""
    n = past.shape[0]

    class Loader:
        epoch = 0

        def __len__(self):
            return (n // batch_size) // world

        def __iter__(self):
            Loader.epoch += 1
            perm = np.random.default_rng(seed + Loader.epoch).permutation(n)
            nb = (n // batch_size) // world * world
            for b in range(rank, nb, world):
                idx = perm[b * batch_size:(b + 1) * batch_size]
                yield past[idx], future[idx]

    return Loader()


def main():
    ap = argparse.ArgumentParser(description="Train a crowd-macroprops DDPM-UNet (MI355X-native path).")
    ap.add_argument('--config-yml-file', type=str, default='config/ATC.yml')
    ap.add_argument('--configList-yml-file', type=str, default=None)
    ap.add_argument('--arch', type=str, default='DDPM-UNet')
    ap.add_argument('--baseline-ckpt', type=str, default=None, help='Baseline model path')
    ap.add_argument('--data-npy', type=str, default=None, help='training sequences [N,C,ROWS,COLS,T] (.npy)')
    ap.add_argument('--synthetic-samples', type=int, default=1024)
    ap.add_argument('--epochs', type=int, default=None, help='override TRAIN.EPOCHS')
    ap.add_argument('--device', type=int, default=None)
    args = ap.parse_args()
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")
    if args.arch not in ("DDPM-UNet", "FM-UNet"):
        raise SystemExit(f"{args.arch}: only the UNet-backbone generators (DDPM-UNet, FM-UNet) are implemented on this path")
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    device = args.device if args.device is not None else int(os.environ.get("LOCAL_RANK", 0))

    from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model
    from crowdmod_ddpm_4d_amd.flow_matching import FM_model
    from generate_samples import windows
    cfg = cfgmod.getYamlConfig(args.config_yml_file, args.configList_yml_file)
    res = cfgmod.resolve(cfg, args.arch)
    if args.epochs is not None and res.train is not None:
        res.train["EPOCHS"] = int(args.epochs)
    mprops = 3  # train.py:61 of the reference
    model = (FM_model if args.arch == "FM-UNet" else DDPM_model)(cfg, args.arch, mprops, device=device)
    if args.epochs is not None:
        model.res = res
    nparams = sum(int(np.prod(v.shape)) for k, v in model.denoiser.state_dict().items()
                  if k != "time_embeddings.time_blocks.0.weight")
    logging.info("Total trainable parameters at denoiser:%d", nparams)
    if args.data_npy:
        seq = np.load(args.data_npy).astype(np.float32)
        past, fut = windows(seq, res.past_len, res.future_len, stride=1, mprops=mprops)
    else:
        n = args.synthetic_samples
        sp = (n, mprops, res.rows, res.cols, res.past_len)
        sf = (n, mprops, res.rows, res.cols, res.future_len)
        past = prng.normal(11, "train/past", int(np.prod(sp))).reshape(sp)
        fut = prng.normal(11, "train/future", int(np.prod(sf))).reshape(sf)
    loader = make_loader(past, fut, res.batch_size, seed=42, rank=rank, world=world)
    logging.info("=======>>>> Init training for %s dataset with %s architecture: %d windows, %d batches/epoch/rank",
                 cfg.DATASET.get("NAME", "?"), args.arch, past.shape[0], len(loader))

    grad_sync = loss_sync = None
    if world > 1:
        from crowdmod_ddpm_4d_amd import distributed as cdist
        cdist.init_process_group()
        grad_sync = cdist.GradAverager()
        loss_sync = cdist.mean_over_ranks
        model.set_data_parallel(rank, world)
    save_dir = cfg.DATA_FS.SAVE_DIR
    os.makedirs(save_dir, exist_ok=True)
    logf = open(os.path.join(save_dir, "train_log.jsonl"), "a") if rank == 0 else None

    def log(rec):
        logging.info("epoch %d: train_loss %.5f lr %.3g", rec["epoch"], rec["train_loss"], rec["lr"])
        if logf:
            logf.write(json.dumps(rec) + "\n")
            logf.flush()

    model.train(loader, args.baseline_ckpt, log=log, grad_sync=grad_sync, save=(rank == 0), loss_sync=loss_sync)
    logging.info("Trained model %s saved in %s", args.arch, save_dir)


if __name__ == '__main__':
    main()
