"""Error of an fp32 dot product formed from exact three-way bf16 splits (six cross terms, fp32 accumulate) against the plain
fp32 chain, both measured against fp64 -- the arithmetic of the six-term kernels (cm_conv_ups.hip, cm_conv_wino.hip,
cm_conv_qr.hip).  numpy only; the matrix instruction's internal order is modelled as a k-ordered fp32 accumulation of the
exact bf16 x bf16 products (each fits fp32 exactly: 8 x 8 mantissa bits)."""
import numpy as np


def bf16_rne(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    hi = bf16_rne(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16_rne(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = bf16_rne(r2)
    return hi, mid, lo


def split2_f16(x):
    """h2: hi = RNE_f16(x), mid = RNE_f16(x - hi) (numpy's float16 conversion rounds to nearest even, subnormals kept)."""
    hi = x.astype(np.float16).astype(np.float32)
    mid = (x - hi).astype(np.float32).astype(np.float16).astype(np.float32)
    return hi, mid


def h2_vs_six(K=864, N=4096, seed=0):
    """The default plan's h2 form (f16 two-way splits, three cross terms) next to the six-term bf16 form on the operand
    distribution of a Winograd layer: SiLU of a standard normal against weights 0.05 N(0, 1), K = 27 taps x 32 channels."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((N, K)).astype(np.float32)
    a = (a / (1 + np.exp(-a))).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    ref = (a.astype(np.float64) * w.astype(np.float64)).sum(1)
    scale = np.sqrt((ref ** 2).mean())

    def chain(pairs):
        acc = np.zeros(N, np.float32)
        for k0 in range(0, K, 16):
            for A, W in pairs:
                for k in range(k0, k0 + 16):
                    acc = (acc.astype(np.float64) + A[:, k].astype(np.float64) * W[:, k].astype(np.float64)).astype(np.float32)
        return acc
    A3, W3 = split3(a), split3(w)
    Ah, Am = split2_f16(a)
    Wh, Wm = split2_f16(w)
    Wsh, Wsm = split2_f16(w * 2.0 ** 13)                   # weights as the plan packs them: w * 2^k, accumulators times 2^-k
    forms = {
        "fp32 chain": chain([(a, w)]),
        "six-term bf16": chain([(A3[0], W3[2]), (A3[2], W3[0]), (A3[1], W3[1]), (A3[0], W3[1]), (A3[1], W3[0]), (A3[0], W3[0])]),
        "three-term bf16": chain([(A3[0], W3[1]), (A3[1], W3[0]), (A3[0], W3[0])]),
        "h2 (f16 x 2, 3 terms)": chain([(Ah, Wm), (Am, Wh), (Ah, Wh)]),
        "h2, weights x 2^13": chain([(Ah, Wsm), (Am, Wsh), (Ah, Wsh)]) * np.float32(2.0 ** -13),
    }
    out = {}
    for name, v in forms.items():
        e = v.astype(np.float64) - ref
        out[name] = (np.sqrt((e ** 2).mean()) / scale, np.abs(e).max() / scale)
        print("%-24s rms error / rms value %.2e   max %.2e" % (name, out[name][0], out[name][1]))
    return out


def main(K=512, N=4096, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((N, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    ref = (a.astype(np.float64) * w.astype(np.float64)).sum(1)
    acc = np.zeros(N, np.float32)
    for k in range(K):                                     # plain fp32 chain (fma per k)
        acc = (acc.astype(np.float64) + a[:, k].astype(np.float64) * w[:, k].astype(np.float64)).astype(np.float32)
    A, W = split3(a), split3(w)
    assert np.array_equal((A[0].astype(np.float64) + A[1] + A[2]).astype(np.float32), a)   # the split is exact
    terms = [(0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)]                                # small products first
    acc6 = np.zeros(N, np.float32)
    for k0 in range(0, K, 16):                             # one matrix instruction = 16 channels of one term
        for ta, tb in terms:
            for k in range(k0, k0 + 16):
                acc6 = (acc6.astype(np.float64) + A[ta][:, k].astype(np.float64) * W[tb][:, k].astype(np.float64)).astype(np.float32)
    acc3 = np.zeros(N, np.float32)
    for k0 in range(0, K, 16):
        for ta, tb in terms[3:]:
            for k in range(k0, k0 + 16):
                acc3 = (acc3.astype(np.float64) + A[ta][:, k].astype(np.float64) * W[tb][:, k].astype(np.float64)).astype(np.float32)
    scale = np.sqrt((ref ** 2).mean())
    for name, v in (("fp32 chain", acc), ("six-term bf16", acc6), ("three-term bf16", acc3)):
        e = v.astype(np.float64) - ref
        print("%-16s rms error / rms value %.2e   max %.2e" % (name, np.sqrt((e ** 2).mean()) / scale, np.abs(e).max() / scale))


if __name__ == "__main__":
    main()
    print("-- h2 against the six-term form (K = 864) --")
    h2_vs_six()
