"""Sampling metrics with the per-frame reductions on the device.

Mirror of the reduction-type metrics of the reference's `MetricsGenerator`
(/root/reference/utils/metrics/metricsGenerator.py): PSNR / masked PSNR (+ MAX over the repeats of a past
sequence, + per-frame tables), relative density error (+ MIN), total variation over time.  The reference loops
over every sample and frame in Python on `.cpu()` copies; here one kernel (cm_frame_metrics) reduces
`[N,C,H,W,F]` to `[N,C,F]` partial sums and only those come back.  SSIM and the energy metric run on the host
(scipy / numpy), as the reference's own do (skimage / torch CPU), and so do the motion-feature histogram metrics
(MF_MSE, MF_BHATT: numpy, vectorised over all volumes)."""
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


# ---- motion-feature histograms (utils/metrics/motionFeatureExtractor.py) ---------------------------------------------------
def _mf_polar(seq):
    """MotionFeatureExtractor.compute_norm_angle_4samples + mag_rho_transform (motionFeatureExtractor.py:34-59) for all
    sequences at once.  seq: [N, C, r, c, F] (channel 1 = v_x, 2 = v_y) -> (log-magnitude, angle), each [N, F, r, c] float64.
    The magnitude and the angle are formed in float32 (the reference's tensors) and widened; the magnitude is min-max scaled
    to [0, 255] PER GRID CELL over the F frames of its sequence (sklearn's MinMaxScaler sees an [F, r c] matrix: one feature
    per cell; a cell whose magnitude does not change maps to 0) and compressed with log2(1 + .) to [0, 8]."""
    seq = np.asarray(seq, dtype=np.float32)
    vx, vy = np.moveaxis(seq[:, 1], -1, 1), np.moveaxis(seq[:, 2], -1, 1)          # [N, F, r, c]
    mag = np.sqrt(vx * vx + vy * vy).astype(np.float64)
    ang = np.arctan2(vy, vx).astype(np.float64)
    lo, hi = mag.min(axis=1, keepdims=True), mag.max(axis=1, keepdims=True)
    rng = hi - lo
    rng = np.where(rng < 10 * np.finfo(np.float64).eps, 1.0, rng)                  # sklearn _handle_zeros_in_scale
    scale = 255.0 / rng
    return np.log2(mag * scale + (0.0 - lo * scale) + 1.0), ang


def _mf_bins(x, lo, hi, nbins):
    """Bin index as np.histogram2d / np.digitize(x, np.linspace(lo, hi, nbins + 1)) - 1 assign it: right-open bins found with
    searchsorted(side='right'); -1 / nbins = outside.  `closed`: histogram2d also takes x == hi into the last bin."""
    edges = np.linspace(lo, hi, nbins + 1)
    return np.searchsorted(edges, x, side="right") - 1, edges


def motion_feature_vectors(seq, f, k, gamma, num_magnitude_bins=16, num_angle_bins=16):
    """The two motion-feature vectors of every sequence as get_motion_feature_2D_hist / get_motion_feature_1D_hist build
    them (motionFeatureExtractor.py:166-284): the sequence is cut into volumes of f frames x k x k cells (ragged at the far
    edges); per volume
      2-D: counts over (log-magnitude in [0, 8], angle in [-pi, pi]) with 16 x 16 bins, the whole first magnitude bin moved to
           the angle bin 8 ("angle 0" for motionless cells);
      1-D: per angle bin the sum of log-magnitude ** gamma;
    the volumes' histograms are concatenated in (frame block, row block, column block) order and divided by (their sum + 1).
    One pass of integer bin indices + np.add.at instead of the reference's histogram call per volume.
    Returns (mf2d [N, nvol * 256], mf1d [N, nvol * 16])."""
    mag, ang = _mf_polar(seq)
    N, F, r, c = mag.shape
    nm, na = int(num_magnitude_bins), int(num_angle_bins)
    nvf, nvr, nvc = -(-F // f), -(-r // k), -(-c // k)
    vol = ((np.arange(F) // f)[:, None, None] * nvr + (np.arange(r) // k)[None, :, None]) * nvc + (np.arange(c) // k)[None, None, :]
    vol = np.broadcast_to(vol[None], mag.shape)
    nvol = nvf * nvr * nvc
    mb, _ = _mf_bins(mag, 0.0, 8.0, nm)
    ab, _ = _mf_bins(ang, -np.pi, np.pi, na)
    samp = np.broadcast_to(np.arange(N)[:, None, None, None], mag.shape)
    # 2-D counts: histogram2d keeps x == right edge in the last bin and drops everything else outside the range
    mb2 = np.where(mag == 8.0, nm - 1, mb)
    ab2 = np.where(ang == np.pi, na - 1, ab)
    ok = (mb2 >= 0) & (mb2 < nm) & (ab2 >= 0) & (ab2 < na)
    h2 = np.zeros((N, nvol, nm, na))
    np.add.at(h2, (samp[ok], vol[ok], mb2[ok], ab2[ok]), 1.0)
    first = h2[:, :, 0, :].sum(axis=2)                       # set_zero_angle_to_smallMag
    h2[:, :, 0, :] = 0.0
    h2[:, :, 0, na // 2] = first
    mf2 = h2.reshape(N, -1)
    mf2 = mf2 / (mf2.sum(axis=1, keepdims=True) + 1.0)
    # 1-D weighted sums: np.digitize has no closed last bin (an angle of exactly +pi is dropped)
    ok1 = (ab >= 0) & (ab < na)
    h1 = np.zeros((N, nvol, na))
    np.add.at(h1, (samp[ok1], vol[ok1], ab[ok1]), np.power(mag[ok1], gamma))
    mf1 = h1.reshape(N, -1)
    mf1 = mf1 / (mf1.sum(axis=1, keepdims=True) + 1.0)
    return mf2, mf1


def bhattacharyya(P, Q, epsilon=1e-2):
    """get_bhattacharyya_dist_coef (motionFeatureExtractor.py:286-303) row-wise: coefficient sum sqrt(P Q) clipped to
    [epsilon, 1], distance -log(coefficient)."""
    coef = np.clip(np.sqrt(np.asarray(P) * np.asarray(Q)).sum(axis=-1), epsilon, 1.0)
    return -np.log(coef), coef


def motion_feature_tables(pred, gt, f, k, gamma, mse_metric=True, bhatt_metrics=True):
    """MetricsGenerator.compute_motion_feature_metrics (metricsGenerator.py:93-112, 240-258): per sample the (2-D based,
    1-D based) mean squared error / Bhattacharyya distance and coefficient between the ground-truth and predicted vectors."""
    p2, p1 = motion_feature_vectors(pred, f, k, gamma)
    g2, g1 = motion_feature_vectors(gt, f, k, gamma)
    out = {"MF_MSE": None, "MF_BHATT_DIST": None, "MF_BHATT_COEF": None}
    if mse_metric:
        out["MF_MSE"] = np.stack([((g2 - p2) ** 2).mean(axis=1), ((g1 - p1) ** 2).mean(axis=1)], axis=1)
    if bhatt_metrics:
        d2, c2 = bhattacharyya(g2, p2)
        d1, c1 = bhattacharyya(g1, p1)
        out["MF_BHATT_DIST"] = np.stack([d2, d1], axis=1)
        out["MF_BHATT_COEF"] = np.stack([c2, c1], axis=1)
    return out


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

    def compute_motion_feature_metrics(self, mse_metric=False, bhatt_metrics=False, f=1, k=4, gamma=0.5):
        """metricsGenerator.py:240-258, on the host (numpy), like the reference's; f / k / gamma = METRICS.MOTION_FEATURE."""
        pred, gt = self._pred_gt
        self.data_dict.update(motion_feature_tables(pred, gt, int(f), int(k), float(gamma), mse_metric, bhatt_metrics))

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
                "ENERGY": "GT,PRED", "MIN-ENERGY": "GT,PRED", "MF_MSE": "MSE_Hist_2D_Based,MSE_Hist_1D_Based",
                "MF_BHATT_DIST": "BHATT_DIST_Hist_2D_Based,BHATT_DIST_Hist_1D_Based",
                "MF_BHATT_COEF": "BHATT_COEF_Hist_2D_Based,BHATT_COEF_Hist_1D_Based"}

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


def compute_metrics(mg: MetricsGenerator, metric: str, chunkRepdPastSeq: int, eps: float, motion_feature=None):
    """utils/metrics/metricsGenerator.py:379-395.  `motion_feature`: the METRICS.MOTION_FEATURE mapping (f, k, GAMMA) of the
    config.  The reference's command line advertises MOTION_FEAT_BHATT (generate_metrics.py:74) while its dispatcher tests
    for MF_BHATT: both spellings select the Bhattacharyya tables here."""
    if metric == "MOTION_FEAT_BHATT":
        metric = "MF_BHATT"
    known = ("PSNR", "MASK_PSNR", "SSIM", "MF_MSE", "MF_BHATT", "ENERGY", "RE_DENSITY", "TV", "ALL")
    if metric not in known:
        raise ValueError(f"metric {metric!r}: this path computes {known}")
    if metric in ("PSNR", "ALL"):
        mg.compute_psnr_metric(chunkRepdPastSeq, eps)
    if metric in ("MASK_PSNR", "ALL"):
        mg.compute_psnr_metric(chunkRepdPastSeq, eps, masked_flag=True)
    if metric in ("SSIM", "ALL"):
        mg.compute_ssim_metric(chunkRepdPastSeq)
    if metric in ("MF_MSE", "MF_BHATT", "ALL"):
        mf = dict(motion_feature or {})
        mg.compute_motion_feature_metrics(metric in ("MF_MSE", "ALL"), metric in ("MF_BHATT", "ALL"),
                                          f=mf.get("f", 1), k=mf.get("k", 4), gamma=mf.get("GAMMA", 0.5))
    if metric == "ENERGY":                      # (the reference lists it under the misspelt 'ALLA': never part of ALL)
        mg.compute_energy_metric(chunkRepdPastSeq)
    if metric in ("RE_DENSITY", "ALL"):
        mg.compute_re_density_metric(chunkRepdPastSeq, eps)
    if metric in ("TV", "ALL"):
        mg.compute_tv_metric()
    return mg
