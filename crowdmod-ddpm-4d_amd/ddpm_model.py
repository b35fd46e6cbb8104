"""Host-side mirror of the reference's `DDPM_model` driver for the DDPM-UNet path
(/root/reference/models/diffusion/ddpm.py:40-108,206-282).

`_generate_ddpm` / `_generate_ddim` keep the reference signatures and return
values -- `(x0, [x_T, ..., x0])` -- but the whole reverse loop runs on the device
behind ONE native call (cm_sample_loop): T sequential UNet forwards + sampler
updates, no host round trip per step.

Noise: by default x_T and z_t come from the device Philox stream keyed by
(seed, global sample index, step) -- results do not depend on how the batch is
sharded over GPUs.  `x_T=` / `noise=` inject explicit tensors (parity tests).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import config as cfgmod
from . import native
from .diffusion import DDPM
from .unet import UNet, _is_torch


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', threshold=1e-4 relative, cooldown 0) as the
    reference configures it (ddpm.py:58-63): after `patience` epochs without a relative improvement of
    1e-4 the rate is multiplied by `factor`, never below `min_lr`."""

    def __init__(self, lr, factor=0.5, patience=10, min_lr=0.0, threshold=1e-4, eps=1e-8):
        self.lr, self.factor, self.patience, self.min_lr = float(lr), float(factor), int(patience), float(min_lr)
        self.threshold, self.eps = float(threshold), float(eps)
        self.best, self.bad = float("inf"), 0

    def step(self, metric: float) -> float:
        m = float(metric)
        if m < self.best * (1.0 - self.threshold):
            self.best, self.bad = m, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            new_lr = max(self.lr * self.factor, self.min_lr)
            if self.lr - new_lr > self.eps:
                self.lr = new_lr
            self.bad = 0
        return self.lr


class DDPM_model:
    _ARCHS = ("DDPM-UNet",)

    def __init__(self, cfg, arch, mprops_count, output_dir=None, from_fixed_past=False, *, device: int = 0,
                 seed: int = 42):
        self.cfg = cfg
        self.arch = arch
        self.mprops_count = int(mprops_count)
        self.output_dir = output_dir
        self.from_fixed_past = from_fixed_past
        self.device = int(device)
        self.seed = int(seed)
        self.denoiser_cfg = self._get_denoiser_cfg()
        self.res = cfgmod.resolve(cfg, arch)
        self.denoiser = self._get_denoiser()
        self._sample_calls = 0
        self.dp_rank, self.dp_world = 0, 1   # data-parallel training position (set_data_parallel)

    def set_data_parallel(self, rank: int, world: int):
        """Data-parallel training (one process per GPU): rank r draws its own timesteps and -- through the
        global sample index rank * batch + b -- its own eps and Dropout3d masks; epoch-level decisions
        (ReduceLROnPlateau, NaN stop, best-loss checkpoint) are taken on the loss averaged over ranks, so every
        replica cuts the rate and stops in the same epoch."""
        if not (0 <= int(rank) < int(world)):
            raise ValueError(f"rank {rank} outside world of {world}")
        self.dp_rank, self.dp_world = int(rank), int(world)
        return self

    def _get_denoiser_cfg(self):
        """ddpm.py:65-72; tolerant of the older schema generations."""
        gen_key, back_key = self.arch.upper().split("-")
        gen = self.cfg.MODEL.get(gen_key) if "MODEL" in self.cfg else None
        if gen is not None and back_key in gen:
            return gen[back_key]
        return gen if gen is not None else self.cfg.get("MODEL")

    def _get_denoiser(self):
        """ddpm.py:74-108."""
        if self.arch not in self._ARCHS:
            raise ValueError(f"Unknown Architecture {self.arch}")
        r = self.res
        return UNet(input_channels=self.mprops_count, output_channels=self.mprops_count,
                    num_res_blocks=r.num_res_blocks, base_channels=r.base_ch,
                    base_channels_multiples=r.base_ch_mult, apply_attention=r.apply_attention,
                    dropout_rate=r.dropout_rate, time_multiple=r.time_emb_mult, condition=r.condition,
                    device=self.device, max_batch=max(1, r.batch_size), seed=self.seed)

    # ------------------------------------------------------------------------------
    def _opts(self, sampler: int, divider: int = 1, first_steps: int = 0, sample_id_base: int = 0,
              seed: Optional[int] = None) -> native.cm_sample_opts:
        r = self.res
        o = native.cm_sample_opts()
        o.sampler = sampler
        o.guidance = native.GUIDANCE_SPARSITY if r.guidance == "Sparsity" else native.GUIDANCE_NONE  # case-sensitive, ddpm.py:223
        if r.guidance == "mass_preservation":
            raise NotImplementedError("mass_preservation guidance is out of scope (O(N^2) finite differences per step)")
        o.lambda_guidance = float(r.lambda_guidance)
        o.ddim_sigma = float(r.sigma)
        o.ddim_divider = int(divider)
        o.first_steps = int(first_steps)
        o.seed = int(self.seed if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        o.sample_id_base = int(sample_id_base)
        o.use_graph = 1 if os.environ.get("CM_USE_GRAPH") else 0   # hipGraph replay of the step (opt-in)
        return o

    def _run_loop(self, past, sampler_obj: DDPM, nsamples: int, opts: native.cm_sample_opts, history: bool,
                  x_T=None, noise=None):
        r = self.res
        L = native.lib()
        B = int(nsamples)
        if int(past.shape[0]) != B:
            raise ValueError(f"past has batch {past.shape[0]}, nsamples={B}")
        C_, H, W, P, F = self.mprops_count, r.rows, r.cols, r.past_len, r.future_len
        h = self.denoiser.eval().ensure(H, W, P, F, B)
        ns = C.c_int32()
        native.check(L.cm_sample_num_steps(sampler_obj._handle, C.byref(opts), C.byref(ns)))
        nsteps = ns.value
        shape = (B, C_, H, W, F)
        if _is_torch(past):
            import torch
            dev = past.device
            pst = past.contiguous().float()
            out = torch.empty(shape, device=dev, dtype=torch.float32)
            hist = torch.empty((nsteps + 1,) + shape, device=dev, dtype=torch.float32) if history else None
            xt = x_T.contiguous().float() if x_T is not None else None
            nz = noise.contiguous().float() if noise is not None else None
            torch.cuda.current_stream(dev).synchronize()   # inputs ready before the library's stream reads them
            native.check(L.cm_sample_loop(h, sampler_obj._handle, pst.data_ptr(), xt.data_ptr() if xt is not None else None,
                                          nz.data_ptr() if nz is not None else None, C.byref(opts), out.data_ptr(),
                                          hist.data_ptr() if hist is not None else None, B, None))
            native.check(L.cm_device_synchronize(self.device))  # results complete on return
            if history:
                return out, [hist[i] for i in range(nsteps + 1)]
            return out, None
        pst = np.ascontiguousarray(past, dtype=np.float32)
        out = np.empty(shape, dtype=np.float32)
        hist = np.empty((nsteps + 1,) + shape, dtype=np.float32) if history else None
        xt = np.ascontiguousarray(x_T, dtype=np.float32) if x_T is not None else None
        nz = np.ascontiguousarray(noise, dtype=np.float32) if noise is not None else None
        if nz is not None and nz.shape[0] < (nsteps - (1 if opts.sampler == native.SAMPLER_DDPM else 0)):
            raise ValueError("noise tensor has too few steps")
        if nz is not None and nz.shape[0] < nsteps:  # DDPM: the t=0 row is never read; pad for the upload
            nz = np.concatenate([nz, np.zeros((nsteps - nz.shape[0],) + nz.shape[1:], np.float32)])
        native.check(L.cm_sample_loop_host(h, sampler_obj._handle, pst.ctypes.data,
                                           xt.ctypes.data if xt is not None else None,
                                           nz.ctypes.data if nz is not None else None, C.byref(opts), out.ctypes.data,
                                           hist.ctypes.data if hist is not None else None, B))
        if history:
            return out, [hist[i] for i in range(nsteps + 1)]
        return out, None

    def _generate_ddpm(self, past, backward_sampler: DDPM, nsamples, history=False, *, x_T=None, noise=None,
                       sample_id_base: int = 0, first_steps: int = 0):
        """ddpm.py:206-236.  Returns (x0, [x_T, (x after every step if history), x0])."""
        opts = self._opts(native.SAMPLER_DDPM, first_steps=first_steps, sample_id_base=sample_id_base,
                          seed=self.seed + 7919 * self._sample_calls)
        self._sample_calls += 1
        x, hist = self._run_loop(past, backward_sampler, nsamples, opts, bool(history), x_T, noise)
        if history:
            return x, hist
        # without history the device-drawn x_T is not copied back (None stands in for it)
        return x, [x_T, x]

    def _generate_ddim(self, past, taus: Sequence[int], backward_sampler: DDPM, nsamples, history=False, *, x_T=None,
                       noise=None, sample_id_base: int = 0):
        """ddpm.py:238-282.  `taus` must be np.arange(0, T-1, divider) as built at ddpm.py:326."""
        taus = np.asarray(taus)
        divider = int(taus[1] - taus[0]) if len(taus) > 1 else max(1, backward_sampler.timesteps)
        expect = np.arange(0, backward_sampler.timesteps - 1, divider)
        if len(taus) != len(expect) or np.any(taus != expect):
            raise ValueError("taus must equal np.arange(0, timesteps-1, divider)")
        opts = self._opts(native.SAMPLER_DDIM, divider=divider, sample_id_base=sample_id_base,
                          seed=self.seed + 7919 * self._sample_calls)
        self._sample_calls += 1
        x, hist = self._run_loop(past, backward_sampler, nsamples, opts, bool(history), x_T, noise)
        if history:
            return x, hist
        # without history the device-drawn x_T is not copied back (None stands in for it)
        return x, [x_T, x]

    # ------------------------------------------------------------------------------
    def _train_step(self, future, past, forward_sampler: DDPM, *, t=None, noise=None, drop_masks=None,
                    rng: Optional[np.random.Generator] = None):
        """ddpm.py:111-121 as the reference defines it (loss only): t ~ U{0..T-1}, (x_t, eps) = q_sample(future, t),
        eps_hat = UNet(x_t, t, past) with Dropout3d active, loss = mse(eps_hat, eps).  Returns (loss, eps_hat).
        The backward pass and the Adam update of ddpm.py:142-144 run inside `_train_one_epoch` (one native
        cm_train_step per batch); this method is the stand-alone loss evaluation with injectable t / noise /
        masks that the parity tests use."""
        future = np.ascontiguousarray(future, dtype=np.float32)
        B = future.shape[0]
        if t is None:
            rng = rng or np.random.default_rng(self.seed)
            t = rng.integers(0, forward_sampler.timesteps, size=(B,))
        x_t, eps = forward_sampler(future, t, noise=noise)
        self.denoiser.train()
        try:
            pred = self.denoiser.forward_train(x_t, t, past, drop_masks=drop_masks, seed=self.seed)
        finally:
            self.denoiser.eval()
        return self.denoiser.mse_loss(pred, eps), pred

    # -- training (ddpm.py:123-202) ---------------------------------------------------
    def _solver(self):
        tr = self.res.train or {}
        sol = tr.get("SOLVER", {}) if hasattr(tr, "get") else {}
        sch = sol.get("SCHEDULER", {}) if hasattr(sol, "get") else {}
        return {
            "epochs": int(tr.get("EPOCHS", 1)) if hasattr(tr, "get") else 1,
            "lr": float(sol.get("LR", 5e-5)), "betas": tuple(float(b) for b in sol.get("BETAS", (0.5, 0.999))),
            "weight_decay": float(sol.get("WEIGHT_DECAY", 0.0)),
            "factor": float(sch.get("FACTOR", 0.5)), "patience": int(sch.get("PATIENCE", 10)),
            "min_lr": float(sch.get("MIN_LR", 0.0)),
        }

    def _ensure_training(self, past, future):
        net = self.denoiser
        B, _, H, W, F = future.shape
        net.ensure(H, W, int(past.shape[4]), F, B)
        if self.dp_world > 1 and getattr(net, "_sample_base", None) != self.dp_rank * B:
            net.set_sample_base(self.dp_rank * B)
        if not getattr(net, "_train_ready", False):
            s = self._solver()
            net.train_init(lr=s["lr"], betas=s["betas"], weight_decay=s["weight_decay"])
            self._lr = s["lr"]
            self._plateau = ReduceLROnPlateau(s["lr"], s["factor"], s["patience"], s["min_lr"])

    def _train_one_epoch(self, forward_sampler: DDPM, loader, epoch, *, rng: Optional[np.random.Generator] = None,
                         grad_sync=None):
        """ddpm.py:123-154: for every (past, future) batch: t ~ U{0..T-1}, q-sample with device-drawn
        noise, train-mode forward, MSE, backward, Adam -- one native call per batch.  `grad_sync`, when
        given, is called between backward and update (data-parallel gradient averaging).
        Returns the mean batch loss (MeanMetric)."""
        # one stream of timesteps per (seed, epoch, rank): replicas must not train on identical draws
        rng = rng or np.random.default_rng([self.seed + epoch, self.dp_rank] if self.dp_world > 1 else self.seed + epoch)
        total, count = 0.0, 0
        for past, future in loader:
            past = np.ascontiguousarray(past, dtype=np.float32)
            future = np.ascontiguousarray(future, dtype=np.float32)
            self._ensure_training(past, future)
            t = rng.integers(0, forward_sampler.timesteps, size=(future.shape[0],)).astype(np.int64)
            self._train_calls = getattr(self, "_train_calls", 0) + 1
            if grad_sync is None:
                loss = self.denoiser.train_step(forward_sampler._handle, future, past, t, None,
                                                seed=self.seed + self._train_calls, apply_update=True)
            else:
                loss = self.denoiser.train_step(forward_sampler._handle, future, past, t, None,
                                                seed=self.seed + self._train_calls, apply_update=False)
                grad_sync(self.denoiser)
                self.denoiser.apply_update()
            total += loss
            count += 1
        return total / max(count, 1)

    def checkpoint_path(self, epoch_tag) -> str:
        """utils/utils.py:120-138: SAVE_DIR + NAME.format(arch, EPOCHS, PAST_LEN, FUTURE_LEN, tag, 'NA')."""
        import os
        name = self.cfg.MODEL.NAME.format(self.arch, self._solver()["epochs"], self.res.past_len, self.res.future_len,
                                          epoch_tag, "NA")
        return os.path.join(self.cfg.DATA_FS.SAVE_DIR, name)

    def save_checkpoint(self, epoch_tag, path: Optional[str] = None) -> str:
        """utils/utils.py:140-147: {"opt": optimizer.state_dict(), "model": model.state_dict()} as a torch
        zip checkpoint the reference reads with torch.load(path, weights_only=True)."""
        import os
        from . import checkpoint
        s = self._solver()
        net = self.denoiser
        net.sync_trained()
        path = path or self.checkpoint_path(epoch_tag)
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        opt = net.optimizer_state_dict(getattr(self, "_lr", s["lr"]), s["betas"], 1e-8, s["weight_decay"])
        checkpoint.save_checkpoint(net.state_dict(), path, opt_state=opt)
        return path

    def train(self, batched_train_data, baseline_ckpt=None, *, log=None, grad_sync=None, save=True, loss_sync=None):
        """ddpm.py:156-202: epochs of _train_one_epoch, ReduceLROnPlateau on the epoch loss, NaN early
        stop, best-loss checkpoint tagged "000" and CHECKPOINTS_TO_KEEP random late epochs.
        Data-parallel: `grad_sync` averages the gradients every step and `loss_sync` (float -> float, e.g.
        distributed.mean_over_ranks) averages the epoch loss, so that the scheduler, the NaN stop and the
        checkpoint decisions are identical on every rank (a rank that stopped alone would leave the others
        blocked in the gradient all-reduce)."""
        import logging
        forward_sampler = DDPM(timesteps=max(2, self.res.timesteps), scale=self.res.scale, device=self.device)
        if baseline_ckpt is not None:
            self.load_checkpoint(baseline_ckpt)
            logging.info("Baseline checkpoint loaded successfully.")
        s = self._solver()
        epochs = s["epochs"]
        best_loss, nan_run = 1e6, 0
        keep = getattr(self, "_keep_override", None)
        if keep is None:
            keep = int(self.cfg.MODEL.get("DDPM", {}).get("CHECKPOINTS_TO_KEEP", 0) or 0)
        rng = np.random.default_rng(self.seed)
        to_save = set(int(v) for v in rng.integers(int(epochs * 0.75), epochs + 1, size=keep)) if keep else set()
        history = []
        for epoch in range(1, epochs + 1):
            epoch_loss = self._train_one_epoch(forward_sampler, batched_train_data, epoch, grad_sync=grad_sync)
            if loss_sync is not None:
                epoch_loss = float(loss_sync(epoch_loss))   # NaN on any rank -> NaN everywhere
            history.append(epoch_loss)
            if log:
                log({"train_loss": epoch_loss, "epoch": epoch, "lr": self._lr})
            new_lr = self._plateau.step(epoch_loss)
            if new_lr != self._lr:
                self._lr = new_lr
                self.denoiser.set_lr(new_lr)
            if np.isnan(epoch_loss):
                nan_run += 1
                logging.warning("Epoch %d: loss is NaN (%d consecutive)", epoch, nan_run)
                if nan_run >= 3:
                    logging.error("Loss has been NaN for 3 consecutive epochs; terminating training early.")
                    break
            else:
                nan_run = 0
            if save and epoch_loss < best_loss:
                best_loss = epoch_loss
                self.save_checkpoint("000")
            if save and epoch in to_save:
                logging.info("Epoch %d: in checkpoints_to_keep set, saving model.", epoch)
                self.save_checkpoint(epoch)
        self.denoiser.sync_trained()
        return history

    def load_checkpoint(self, model_fullname: str):
        """ddpm.py:288: load_state_dict(torch.load(path, map_location='cpu', weights_only=True)['model'])."""
        from . import checkpoint
        self.denoiser.load_state_dict(checkpoint.load_model_state(model_fullname))
        return self

    def generate_metrics(self, batched_test_data, chunkRepdPastSeq, metric, batches_to_use, samples_per_batch,
                         model_fullname=None, output_dir=None, *, rng: Optional[np.random.Generator] = None, eps=None):
        """ddpm.py:336-392: per test batch draw `samples_per_batch / chunkRepdPastSeq` past windows, repeat each
        `chunkRepdPastSeq` times (torch.repeat_interleave), sample all `samples_per_batch` chains (NSAMPLES = 1280 =
        64 pasts x 20 repeats in config/ATC.yml) in ONE device loop, and hand predictions / ground truth to the
        MetricsGenerator -- whose per-frame reductions run on the device as well (cm_frame_metrics), so neither the
        1280 samples' Python double loop nor a `.cpu()` round trip per sample remains.  Returns the MetricsGenerator;
        with `output_dir` its tables are written there as CSV + metrics_files.json (utils/metrics/metricsGenerator.py:
        342-358).  metric: PSNR | MASK_PSNR | RE_DENSITY | TV | ALL (SSIM, motion-feature and energy metrics are CPU
        library code on the reference side and out of scope here)."""
        import logging
        from .metrics import MetricsGenerator, compute_metrics
        r = self.res
        if model_fullname is not None:
            self.load_checkpoint(model_fullname)
        sampler = DDPM(timesteps=r.timesteps, scale=r.scale, device=self.device)
        rng = rng or np.random.default_rng(42)
        samples_per_batch, chunk = int(samples_per_batch), int(chunkRepdPastSeq)
        preds, gts, count = [], [], 0
        for past_test, future_test in batched_test_data:
            past_test = np.asarray(past_test, dtype=np.float32)
            future_test = np.asarray(future_test, dtype=np.float32)
            n = past_test.shape[0]
            idx = rng.permutation(n) if n < samples_per_batch else rng.permutation(n)[:samples_per_batch]
            idx = np.repeat(idx, chunk)[:samples_per_batch]            # repeat_interleave, then the first samples_per_batch
            pasts, futures = past_test[idx], future_test[idx]
            nb = len(idx)
            logging.info("Computing sampling on batch %d: %d chains (%d pasts x %d repeats)", count + 1, nb, -(-nb // chunk), chunk)
            if r.sampler == "DDPM":
                x, _ = self._generate_ddpm(pasts, sampler, nb)
                logging.info("L1 norm %.2f using %s guidance", float(np.mean(np.abs(x[:, 0]))), r.guidance)
            elif r.sampler == "DDIM":
                x, _ = self._generate_ddim(pasts, np.arange(0, r.timesteps - 1, r.ddim_divider), sampler, nb)
            else:
                raise ValueError(f"{r.sampler} sampler not supported")
            preds.append(x)
            gts.append(futures)
            count += 1
            if count == int(batches_to_use):
                break
        if not preds:
            raise ValueError("empty test data")
        mg = MetricsGenerator(np.concatenate(preds), np.concatenate(gts), self.mprops_count, device=self.device)
        if eps is None:
            mp = self.cfg.get("MACROPROPS", {}) if hasattr(self.cfg, "get") else {}
            eps = float(mp.get("EPS", 1e-6)) if hasattr(mp, "get") else 1e-6
        mt = self.cfg.get("METRICS", {}) if hasattr(self.cfg, "get") else {}
        mf = mt.get("MOTION_FEATURE", None) if hasattr(mt, "get") else None    # f, k, GAMMA (metricsGenerator.py:244-245)
        compute_metrics(mg, metric, chunk, eps, motion_feature=mf)
        if output_dir:
            title = f"{r.batch_size * chunk * count} samples in total (BS:{r.batch_size}, Rep:{chunk}, TB:{count})-({self.arch})"
            mg.save_data_metrics(output_dir, title, samples_per_batch)
        return mg

    def sampling(self, batched_test_data, plotType=None, model_fullname=None, plotMprop=None, plotPast=None,
                 samePastSeq=False, macropropPlotter=None, *, rng: Optional[np.random.Generator] = None):
        """ddpm.py:284-334 without the matplotlib tail: load the checkpoint, take the first
        batch, draw NSAMPLES4PLOTS past windows (all equal if `samePastSeq`), run the
        configured sampler and return (predictions, past_idx, pasts, futures).  Plotting
        is the caller's business (`macropropPlotter` is accepted for signature parity)."""
        import logging
        r = self.res
        if model_fullname is not None:
            self.load_checkpoint(model_fullname)
        sampler = DDPM(timesteps=r.timesteps, scale=r.scale, device=self.device)
        rng = rng or np.random.default_rng(42)
        for past_test, future_test in batched_test_data:
            past_test = np.asarray(past_test, dtype=np.float32)
            future_test = np.asarray(future_test, dtype=np.float32)
            nsamples = past_test.shape[0] if self.from_fixed_past else min(r.nsamples4plots, past_test.shape[0])
            if self.from_fixed_past:
                idx = np.arange(nsamples)
            else:
                idx = rng.permutation(past_test.shape[0])[:nsamples]
                if samePastSeq:
                    idx[:] = idx[0]
            pasts, futures = past_test[idx], future_test[idx]
            if r.sampler == "DDPM":
                pred, _ = self._generate_ddpm(pasts, sampler, nsamples)
                logging.info("L1 norm %.2f", float(np.mean(np.abs(pred[:, 0]))))
            elif r.sampler == "DDIM":
                taus = np.arange(0, r.timesteps - 1, r.ddim_divider)
                pred, _ = self._generate_ddim(pasts, taus, sampler, nsamples)
            else:
                raise ValueError(f"{r.sampler} sampler not supported")
            return pred, idx, pasts, futures   # the reference breaks after the first batch (ddpm.py:334)
        raise ValueError("empty test data")
