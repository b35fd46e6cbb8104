"""CPU restatement of the reduction-type sampling metrics (TEST INFRASTRUCTURE ONLY -- see oracle/__init__).

Follows /root/reference/utils/metrics/metricsGenerator.py: `_get_mprops_ranges` :44-68, `_my_psnr` :70-79,
`_my_psnr_masked` :81-86, `_compute_tv` :88-92, `compute_psnr_metric` :120-186, `compute_re_density_metric`
:293-317, `compute_tv_metric` :319-339.  Pinned by tests/golden/metrics.npz (tables produced by the reference's
own MetricsGenerator).  pred / gt: [N, C, H, W, F]."""
import numpy as np


def ranges(gt, m=3):
    return [float(gt[:, c].max() - gt[:, c].min()) for c in range(m)]


def psnr_tables(pred, gt, chunk, eps, masked, m=3):
    N, _, _, _, F = pred.shape
    rng = ranges(gt, m)
    over_time = np.zeros((N, m * F))
    avg = np.zeros((N, m))
    for i in range(N):
        for j in range(F):
            mask = gt[i, 0, :, :, j] > 0.00001
            for c in range(m):
                g, p = gt[i, c, :, :, j], pred[i, c, :, :, j]
                with np.errstate(invalid="ignore"):
                    err = np.mean((g[mask] - p[mask]) ** 2, dtype=np.float64) if masked else np.mean((g - p) ** 2, dtype=np.float64)
                err = max(err, eps)   # Python max keeps a NaN first argument, like the reference
                over_time[i, j * m + c] = 20 * np.log10(rng[c]) - 10 * np.log10(err)
        avg[i] = over_time[i].reshape(F, m).sum(axis=0) / F
    nch = N // chunk
    mx = np.stack([avg[k * chunk:(k + 1) * chunk].max(axis=0) for k in range(nch)])
    mxt = np.stack([over_time[k * chunk:(k + 1) * chunk].max(axis=0) for k in range(nch)])
    return avg, mx, over_time, mxt


def re_density(pred, gt, chunk, eps):
    sp, sg = pred[:, 0].sum(axis=(1, 2)), gt[:, 0].sum(axis=(1, 2))
    re = np.abs(sp - sg) / (sg + eps)
    return re, np.stack([re[k * chunk:(k + 1) * chunk].min(axis=0) for k in range(pred.shape[0] // chunk)])


def tv_over_time(pred, gt, m=3):
    N, _, _, _, F = pred.shape

    def tv(f):
        return np.abs(np.diff(f, axis=0)).sum() + np.abs(np.diff(f, axis=1)).sum()
    out = np.zeros((N, m * F))
    for i in range(N):
        for j in range(F):
            for c in range(m):
                out[i, j * m + c] = np.abs(tv(pred[i, c, :, :, j]) - tv(gt[i, c, :, :, j]))
    return out
