"""Host-side mirror of the reference's flow-matching driver on the UNet backbone
(/root/reference/models/flow_matching/flow_matching.py:14-250, arch "FM-UNet").

The velocity predictor is the same native UNet; sampling is the Euler integrator
    x <- x + (1/N) u(x, idx_i, past),  t_i = linspace(0,1,N)[i],  idx_i = clamp(t_i * TIME_MAX_POS).long()
run as ONE device-resident loop (cm_sample_loop with CM_SAMPLER_FM_EULER); the reference maps the
"Heun" integrator name to the Euler routine as well (flow_matching.py:44-47), and so does this class.
Training draws x0 ~ N(0,1), t ~ U(0,1), builds (x_t, u_target) with the Linear or Conic path on the
host (two axpys on [B,C,H,W,F]) and runs the native train-mode forward + MSE + backward + Adam step
(cm_train_step_xt)."""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

from . import native, prng
from .ddpm_model import DDPM_model
from .diffusion import DDPM


class FM_model(DDPM_model):
    _ARCHS = ("FM-UNet",)

    def __init__(self, cfg, arch, mprops_count, output_dir=None, from_fixed_past=False, *, device: int = 0, seed: int = 42):
        super().__init__(cfg, arch, mprops_count, output_dir, from_fixed_past, device=device, seed=seed)
        self.u_predictor_cfg = self.denoiser_cfg
        self.u_predictor = self.denoiser
        fm = cfg.MODEL.FM
        self.time_max_pos = int(fm.get("TIME_MAX_POS", 1000))
        self.w_type = str(fm.get("W_TYPE", "Linear"))
        self.integrator = str(fm.get("INTEGRATOR", "Euler"))
        steps = fm.get("INTEGRATOR_STEPS", {}) or {}
        self.euler_steps = int(steps.get("EULER", 1000))
        self.w_type_fns = {"Linear": self.w_linear, "Conic": self.w_conic}
        self.integrators = {"Euler": self.sampling_with_euler, "Heun": self.sampling_with_euler}  # flow_matching.py:44-47
        self._sched = None
        self._fm_calls = 0

    # -- probability paths (flow_matching.py:90-102) -----------------------------------
    @staticmethod
    def w_linear(x0, x1, t):
        xt = x0 + t * (x1 - x0)
        return xt, x1 - x0

    @staticmethod
    def w_conic(x0, x1, t):
        xt = t * x1 + (1 - t) * x0
        return xt, (x1 - xt) / (1 - t)

    # -- sampling (flow_matching.py:203-224) --------------------------------------------
    def _schedule(self):
        if self._sched is None:
            self._sched = DDPM(timesteps=2, scale=1.0, device=self.device)  # only a handle: FM does not read it
        return self._sched

    def sampling_with_euler(self, past, nsamples, *, x0=None, steps: Optional[int] = None, sample_id_base: int = 0):
        o = self._opts(native.SAMPLER_FM_EULER, sample_id_base=sample_id_base, seed=self.seed + self._sample_calls)
        self._sample_calls += 1
        o.guidance = native.GUIDANCE_NONE
        o.fm_steps = int(steps or self.euler_steps)
        o.fm_time_max_pos = self.time_max_pos
        x, _ = self._run_loop(past, self._schedule(), nsamples, o, False, x_T=x0)
        return x

    # -- training (flow_matching.py:104-157) ---------------------------------------------
    def _train_one_epoch(self, forward_sampler, loader, epoch, *, rng: Optional[np.random.Generator] = None,
                         grad_sync=None, x0=None, t=None):
        return self._train_one_epoch_fm(loader, epoch, rng=rng, grad_sync=grad_sync, x0=x0, t=t)

    def _train_one_epoch_fm(self, loader, epoch, *, rng: Optional[np.random.Generator] = None, grad_sync=None,
                            x0=None, t=None, drop_masks=None):
        try:
            w_fn = self.w_type_fns[self.w_type]
        except KeyError:
            raise ValueError(f"Unsupported W_TYPE '{self.w_type}'. Available: {list(self.w_type_fns.keys())}")
        rng = rng or np.random.default_rng([self.seed + epoch, self.dp_rank] if self.dp_world > 1 else self.seed + epoch)
        total, count = 0.0, 0
        for past, future in loader:
            past = np.ascontiguousarray(past, dtype=np.float32)
            x1 = np.ascontiguousarray(future, dtype=np.float32)
            self._ensure_training(past, x1)
            B = x1.shape[0]
            self._fm_calls += 1
            tag = f"fm/x0/{self._fm_calls}" + (f"/r{self.dp_rank}" if self.dp_world > 1 else "")   # own x0 per rank
            x0b = prng.normal(self.seed, tag, x1.size).reshape(x1.shape) if x0 is None else x0
            tb = rng.random(B, dtype=np.float32) if t is None else np.asarray(t, dtype=np.float32)
            tv = tb.reshape(-1, 1, 1, 1, 1)
            xt, u_target = w_fn(np.asarray(x0b, dtype=np.float32), x1, tv)
            t_idx = (tb * np.float32(self.time_max_pos)).astype(np.int64)   # (t * time_max_pos).long(): truncation
            loss = self.denoiser.train_step_xt(xt.astype(np.float32), past, t_idx, u_target.astype(np.float32),
                                               drop_masks=drop_masks, seed=self.seed + self._fm_calls,
                                               apply_update=grad_sync is None)
            if grad_sync is not None:
                grad_sync(self.denoiser)
                self.denoiser.apply_update()
            total += loss
            count += 1
        return total / max(count, 1)

    def checkpoint_path(self, epoch_tag) -> str:
        """utils/utils.py:130-131: the FM checkpoints carry the W_TYPE where the DDPM ones say "NA"."""
        name = self.cfg.MODEL.NAME.format(self.arch, self._solver()["epochs"], self.res.past_len, self.res.future_len,
                                          epoch_tag, self.w_type)
        return os.path.join(self.cfg.DATA_FS.SAVE_DIR, name)

    def train(self, batched_train_data, baseline_ckpt=None, *, log=None, grad_sync=None, save=True, loss_sync=None):
        keep = int(self.cfg.MODEL.FM.get("CHECKPOINTS_TO_KEEP", 0) or 0)
        self._keep_override = keep
        return super().train(batched_train_data, baseline_ckpt, log=log, grad_sync=grad_sync, save=save,
                             loss_sync=loss_sync)

    # -- sampling entry (flow_matching.py:250-292) ------------------------------------------
    def sampling(self, batched_test_data, plotType=None, model_fullname=None, plotMprop=None, plotPast=None,
                 samePastSeq=False, macropropPlotter=None, *, rng: Optional[np.random.Generator] = None):
        if model_fullname:
            self.load_checkpoint(model_fullname)
        try:
            integrator = self.integrators[self.integrator]
        except KeyError:
            raise ValueError(f"Unsupported INTEGRATOR '{self.integrator}'. Available: {list(self.integrators.keys())}")
        rng = rng or np.random.default_rng(self.seed)
        for past_test, future_test in batched_test_data:
            past_test = np.asarray(past_test, dtype=np.float32)
            future_test = np.asarray(future_test, dtype=np.float32)
            nsamples = past_test.shape[0] if self.from_fixed_past else min(self.res.nsamples4plots, past_test.shape[0])
            idx = np.arange(nsamples) if self.from_fixed_past else rng.permutation(past_test.shape[0])[:nsamples]
            if samePastSeq and not self.from_fixed_past:
                idx[:] = idx[0]
            pred = integrator(past_test[idx], nsamples)
            return pred, idx, past_test[idx], future_test[idx]
        raise ValueError("empty test data")
