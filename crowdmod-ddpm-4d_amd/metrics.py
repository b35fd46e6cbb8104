"""Sampling metrics with the per-frame reductions on the device.

Mirror of the reduction-type metrics of the reference's `MetricsGenerator`
(/root/reference/utils/metrics/metricsGenerator.py): PSNR / masked PSNR (+ MAX over the repeats of a past
sequence, + per-frame tables), relative density error (+ MIN), total variation over time.  The reference loops
over every sample and frame in Python on `.cpu()` copies; here one kernel (cm_frame_metrics) reduces
`[N,C,H,W,F]` to `[N,C,F]` partial sums and only those come back.  SSIM and the energy metric run on the host
(scipy / numpy), as the reference's own do (skimage / torch CPU); the motion-feature histograms are not built."""
from __future__ import annotations

import numpy as np

from . import native


# ---- CPU metrics (the reference computes these on CPU libraries as well; SURVEY 8 f-4 leaves them off the device) -------
def _ssim2d(x, y, data_range, win=7, K1=0.01, K2=0.03):
    """skimage.metrics.structural_similarity with its defaults, as the reference calls it
    (utils/metrics/metricsGenerator.py:206-208: two 2-D float images + data_range): 7x7 uniform window, sample covariance
    (N / (N - 1)), C1 = (K1 R)^2, C2 = (K2 R)^2, mean of the SSIM map with a (win - 1) / 2 border cropped.
    PARITY UNPINNED: scikit-image is not installed in the build container, so no reference output exists for this function;
    it restates the published algorithm (Wang et al. 2004, as implemented by scikit-image 0.2x) and is checked against a
    brute-force evaluation of the same formula (tests/test_host_cpu.py)."""
    from scipy.ndimage import uniform_filter
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if min(x.shape) < win:
        raise ValueError(f"SSIM needs images of at least {win}x{win} (got {x.shape}); the reference raises as well")
    npx = win * win
    cov_norm = npx / (npx - 1.0)
    ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
    uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    pad = (win - 1) // 2
    return float(S[pad:S.shape[0] - pad, pad:S.shape[1] - pad].mean(dtype=np.float64))


def ssim_tables(pred, gt, ranges, chunk, mprops_count=3):
    """metricsGenerator.py:188-238: SSIM per (sample, property) averaged over the frames, per frame, and their maxima over
    each chunk of `chunk` repeats.  pred / gt: [N, C, H, W, F]."""
    pred, gt = np.asarray(pred), np.asarray(gt)
    N, _, _, _, F = pred.shape
    m = mprops_count
    over_time = np.zeros((N, m * F))
    for i in range(N):
        for j in range(F):
            for c in range(m):
                over_time[i, j * m + c] = _ssim2d(gt[i, c, :, :, j], pred[i, c, :, :, j], ranges[c])
    avg = over_time.reshape(N, F, m).sum(axis=1) / F
    nchunk = N // chunk
    mx = np.stack([avg[i * chunk:(i + 1) * chunk].max(axis=0) for i in range(nchunk)]) if nchunk else np.zeros((0, m))
    mxt = np.stack([over_time[i * chunk:(i + 1) * chunk].max(axis=0) for i in range(nchunk)]) if nchunk else np.zeros((0, m * F))
    return {"SSIM": avg, "MAX_SSIM": mx, "SSIM_OVER_TIME": over_time, "MAX_SSIM_OVER_TIME": mxt}


def compute_energy(x, delta_t=0.5, delta_l=1.0):
    """models/guidance.py:10-42 (mass-conservation residual energy), float32 like the reference's torch code.
    x: [B, 3, H, W, L] -> [B]."""
    x = np.asarray(x, dtype=np.float32)
    _, _, H, W, L = x.shape
    it, il = np.float32(1.0 / delta_t), np.float32(1.0 / delta_l)
    rho, vx, vy = x[:, 0], x[:, 1], x[:, 2]
    c = (slice(None), slice(1, -1), slice(1, -1))
    term1 = it * (rho[c + (slice(1, None),)] - rho[c + (slice(None, -1),)])
    t0 = slice(None, -1)
    term2 = il * rho[:, 1:-1, 1:-1, t0] * ((vx[:, 2:, 1:-1, t0] - vx[:, 1:-1, 1:-1, t0]) + (vy[:, 1:-1, 2:, t0] - vy[:, 1:-1, 1:-1, t0]))
    term3 = il * (rho[:, 2:, 1:-1, t0] - rho[:, 1:-1, 1:-1, t0]) * vx[:, 1:-1, 1:-1, t0]
    term4 = il * (rho[:, 1:-1, 2:, t0] - rho[:, 1:-1, 1:-1, t0]) * vy[:, 1:-1, 1:-1, t0]
    f = term1 + term2 + term3 + term4
    e = np.float32(0.5) * np.sum(f * f, axis=(1, 2, 3), dtype=np.float32)
    return e / np.float32(H * W * L)


def energy_tables(pred, gt, chunk, mprops_factor=None):
    """metricsGenerator.py:260-291: energy of the ground-truth and predicted sequences (columns GT, PRED) and the minima over
    each chunk of repeats.  The reference's method reads `mprops_factor` before assigning it (metricsGenerator.py:264) and
    raises UnboundLocalError as written; here the factor is an argument (default: none, i.e. 1)."""
    pred, gt = np.asarray(pred, dtype=np.float32), np.asarray(gt, dtype=np.float32)
    if mprops_factor is not None:
        fac = np.asarray(mprops_factor, dtype=np.float32)[:pred.shape[1], None, None, None]
        pred, gt = pred * fac[None], gt * fac[None]
    e = np.stack([compute_energy(gt, 1, 1), compute_energy(pred, 1, 1)], axis=1).astype(np.float64)
    nchunk = pred.shape[0] // chunk
    mn = np.stack([e[i * chunk:(i + 1) * chunk].min(axis=0) for i in range(nchunk)]) if nchunk else np.zeros((0, 2))
    return {"ENERGY": e, "MIN-ENERGY": mn}


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

    def compute_ssim_metric(self, chunkRepdPastSeq):
        """metricsGenerator.py:188-238, on the host (scipy), like the reference's skimage call."""
        pred, gt = self._pred_gt
        self.data_dict.update(ssim_tables(pred, gt, self.ranges, chunkRepdPastSeq, self.mprops_count))

    def compute_energy_metric(self, chunkRepdPastSeq, mprops_factor=None):
        """metricsGenerator.py:260-291, on the host."""
        pred, gt = self._pred_gt
        self.data_dict.update(energy_tables(pred, gt, chunkRepdPastSeq, mprops_factor))

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
                "RE_DENSITY": frames, "MIN_RE_DENSITY": frames, "TV_OVER_TIME": per_frame,
                "SSIM": flat, "MAX_SSIM": flat, "SSIM_OVER_TIME": per_frame, "MAX_SSIM_OVER_TIME": per_frame,
                "ENERGY": "GT,PRED", "MIN-ENERGY": "GT,PRED"}

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
    known = ("PSNR", "MASK_PSNR", "SSIM", "ENERGY", "RE_DENSITY", "TV", "ALL")
    if metric not in known:
        raise ValueError(f"metric {metric!r}: this path computes {known}; MF_MSE / MF_BHATT (motion-feature histograms, "
                         f"utils/metrics/motionFeatureExtractor.py) are not built")
    if metric in ("PSNR", "ALL"):
        mg.compute_psnr_metric(chunkRepdPastSeq, eps)
    if metric in ("MASK_PSNR", "ALL"):
        mg.compute_psnr_metric(chunkRepdPastSeq, eps, masked_flag=True)
    if metric in ("SSIM", "ALL"):
        mg.compute_ssim_metric(chunkRepdPastSeq)
    if metric == "ENERGY":                      # (the reference lists it under the misspelt 'ALLA': never part of ALL)
        mg.compute_energy_metric(chunkRepdPastSeq)
    if metric in ("RE_DENSITY", "ALL"):
        mg.compute_re_density_metric(chunkRepdPastSeq, eps)
    if metric in ("TV", "ALL"):
        mg.compute_tv_metric()
    return mg
