"""Sampling metrics with the per-frame reductions on the device.

Mirror of the reduction-type metrics of the reference's `MetricsGenerator`
(/root/reference/utils/metrics/metricsGenerator.py): PSNR / masked PSNR (+ MAX over the repeats of a past
sequence, + per-frame tables), relative density error (+ MIN), total variation over time.  The reference loops
over every sample and frame in Python on `.cpu()` copies; here one kernel (cm_frame_metrics) reduces
`[N,C,H,W,F]` to `[N,C,F]` partial sums and only those come back.  SSIM (skimage), motion-feature histograms
and the energy metric stay out of scope (CPU libraries on the reference side)."""
from __future__ import annotations

import numpy as np

from . import native


class MetricsGenerator:
    def __init__(self, pred, gt, mprops_count: int = 3, device: int = 0):
        """pred / gt: arrays [N, C, H, W, F] (or lists of [C,H,W,F] sequences like the reference takes)."""
        pred = np.ascontiguousarray(np.stack([np.asarray(p) for p in pred]) if isinstance(pred, (list, tuple)) else pred,
                                    dtype=np.float32)
        gt = np.ascontiguousarray(np.stack([np.asarray(p) for p in gt]) if isinstance(gt, (list, tuple)) else gt,
                                  dtype=np.float32)
        if pred.shape != gt.shape or pred.ndim != 5:
            raise ValueError(f"pred {pred.shape} / gt {gt.shape}: expected equal [N,C,H,W,F]")
        self.mprops_count = int(mprops_count)
        N, C_, H, W, F = pred.shape
        if C_ < self.mprops_count:
            raise ValueError("fewer channels than MPROPS_COUNT")
        dp = native.DeviceBuffer.from_array(pred, device)
        dg = native.DeviceBuffer.from_array(gt, device)
        out = np.empty((N, C_, F, 8), dtype=np.float64)
        mm = np.empty((N, C_, F, 2), dtype=np.float32)
        native.check(native.lib().cm_frame_metrics(device, dp.ptr, dg.ptr, N, C_, H, W, F, out.ctypes.data, mm.ctypes.data))
        self.N, self.F, self.npix = N, F, H * W
        self._pred_gt = (pred, gt)
        self._red = out
        m = self.mprops_count
        # _get_mprops_ranges (metricsGenerator.py:44-68): global max - global min of the ground truth per property
        self.ranges = [float(mm[:, c, :, 1].max() - mm[:, c, :, 0].min()) for c in range(m)]
        self.rho_range, self.vx_range, self.vy_range = (self.ranges + [0.0, 0.0])[:3]
        self.data_dict = {}

    # metricsGenerator.py:70-86
    @staticmethod
    def _psnr(err, data_range, eps):
        err = np.maximum(err, eps)
        return 20 * np.log10(data_range) - 10 * np.log10(err)

    def compute_psnr_metric(self, chunkRepdPastSeq, eps, masked_flag=False):
        """metricsGenerator.py:120-186: PSNR per (sample, property) averaged over the frames, per frame, and their
        maxima over each chunk of `chunkRepdPastSeq` repeats."""
        m, N, F = self.mprops_count, self.N, self.F
        r = self._red[:, :m]
        if masked_flag:
            with np.errstate(invalid="ignore", divide="ignore"):
                mse = r[..., 1] / r[..., 2]      # empty mask: nan, as np.mean of an empty selection gives
        else:
            mse = r[..., 0] / float(self.npix)
        rng = np.asarray(self.ranges, dtype=np.float64).reshape(1, m, 1)
        with np.errstate(invalid="ignore"):
            frame = self._psnr(mse, rng, eps)                                    # [N, m, F]
        over_time = np.transpose(frame, (0, 2, 1)).reshape(N, F * m)             # rho_f0, vx_f0, vy_f0, rho_f1, ...
        avg = frame.sum(axis=2) / F
        nchunk = N // chunkRepdPastSeq
        mx = np.stack([avg[i * chunkRepdPastSeq:(i + 1) * chunkRepdPastSeq].max(axis=0) for i in range(nchunk)]) if nchunk else np.zeros((0, m))
        mxt = np.stack([over_time[i * chunkRepdPastSeq:(i + 1) * chunkRepdPastSeq].max(axis=0) for i in range(nchunk)]) if nchunk else np.zeros((0, F * m))
        p = "MASK_" if masked_flag else ""
        self.data_dict[f"{p}PSNR"] = avg
        self.data_dict[f"MAX_{p}PSNR"] = mx
        self.data_dict[f"{p}PSNR_OVER_TIME"] = over_time
        self.data_dict[f"MAX_{p}PSNR_OVER_TIME"] = mxt

    def compute_re_density_metric(self, chunkRepdPastSeq, eps):
        """metricsGenerator.py:293-317: |sum(rho_pred) - sum(rho_gt)| / (sum(rho_gt) + eps) per frame."""
        sp, sg = self._red[:, 0, :, 5], self._red[:, 0, :, 6]
        re = np.abs(sp - sg) / (sg + eps)
        nchunk = self.N // chunkRepdPastSeq
        mn = np.stack([re[i * chunkRepdPastSeq:(i + 1) * chunkRepdPastSeq].min(axis=0) for i in range(nchunk)]) if nchunk else np.zeros((0, self.F))
        self.data_dict["RE_DENSITY"] = re
        self.data_dict["MIN_RE_DENSITY"] = mn

    def compute_tv_metric(self):
        """metricsGenerator.py:319-339: |TV(pred) - TV(gt)| per (frame, property)."""
        m = self.mprops_count
        d = np.abs(self._red[:, :m, :, 3] - self._red[:, :m, :, 4])              # [N, m, F]
        self.data_dict["TV_OVER_TIME"] = np.transpose(d, (0, 2, 1)).reshape(self.N, self.F * m)

    # metricsGenerator.py:342-358 (CSV tables + an index JSON; the reference's boxplots are matplotlib code, not mirrored)
    def headers(self):
        names = ["rho", "vx", "vy", "unc"][: self.mprops_count]
        per_frame = ",".join(f"{n}_f{f + 1}" for f in range(self.F) for n in names)
        flat = ",".join(names)
        frames = ",".join(f"f{f + 1}" for f in range(self.F))
        return {"PSNR": flat, "MAX_PSNR": flat, "PSNR_OVER_TIME": per_frame, "MAX_PSNR_OVER_TIME": per_frame,
                "MASK_PSNR": flat, "MAX_MASK_PSNR": flat, "MASK_PSNR_OVER_TIME": per_frame, "MAX_MASK_PSNR_OVER_TIME": per_frame,
                "RE_DENSITY": frames, "MIN_RE_DENSITY": frames, "TV_OVER_TIME": per_frame}

    def save_data_metrics(self, output_dir, title, samples_per_batch):
        import json
        import os
        os.makedirs(output_dir, exist_ok=True)
        index = {"title": title}
        for name, header in self.headers().items():
            data = self.data_dict.get(name)
            if data is None:
                continue
            fn = os.path.join(output_dir, f"{name}_NS{int(samples_per_batch)}.csv")
            np.savetxt(fn, np.asarray(data, dtype=np.float64), delimiter=",", header=header, comments="")
            index[name] = fn
        with open(os.path.join(output_dir, "metrics_files.json"), "w") as fo:
            json.dump(index, fo, indent=2)
        return index


def compute_metrics(mg: MetricsGenerator, metric: str, chunkRepdPastSeq: int, eps: float):
    """utils/metrics/metricsGenerator.py:379-395 for the metrics this path implements."""
    known = ("PSNR", "MASK_PSNR", "RE_DENSITY", "TV", "ALL")
    if metric not in known:
        raise ValueError(f"metric {metric!r}: this path computes {known}; SSIM / MF_* / ENERGY are CPU-library metrics of the reference")
    if metric in ("PSNR", "ALL"):
        mg.compute_psnr_metric(chunkRepdPastSeq, eps)
    if metric in ("MASK_PSNR", "ALL"):
        mg.compute_psnr_metric(chunkRepdPastSeq, eps, masked_flag=True)
    if metric in ("RE_DENSITY", "ALL"):
        mg.compute_re_density_metric(chunkRepdPastSeq, eps)
    if metric in ("TV", "ALL"):
        mg.compute_tv_metric()
    return mg
