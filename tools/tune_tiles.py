"""Offline tile tuner (run on the MI355X box): for every 3x3x3 convolution of the reference layer shapes,
time all admissible tile geometries through cm_debug_time_conv at B = 64 and print the rows of
crowdmod-ddpm-4d_amd/csrc/cm_tuned_tiles.inc.

    python tools/tune_tiles.py [--grids atc,cr120,x2] [--channels 4] > gpurun_out/tuned.txt

The table is keyed by layer shape, never by batch, so the choice (and every rounding that depends on
the tile geometry) is the same however a batch is sharded."""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GRIDS = {"atc": (12, 36), "cr120": (28, 24), "x2": (24, 72)}


def tune_one(grid, channels, nb64, nb128, B, iters):
    """Runs in a child process (the NB policy is read from the environment when the plan is built)."""
    import numpy as np
    from crowdmod_ddpm_4d_amd import native, spec
    from crowdmod_ddpm_4d_amd.unet import UNet
    H, W = GRIDS[grid]
    net = UNet(input_channels=channels, output_channels=channels, num_res_blocks=1, base_channels=32,
               base_channels_multiples=(1, 2, 4), apply_attention=(False, False, True), dropout_rate=0.1,
               time_multiple=4, condition="Past", max_batch=B)
    net.load_state_dict(spec.init_params(net.cfg, 42))
    rng = np.random.default_rng(0)
    fut = rng.standard_normal((B, channels, H, W, 3), dtype=np.float32)
    past = rng.standard_normal((B, channels, H, W, 5), dtype=np.float32)
    net(fut, np.arange(B) * 7 % 1000, past)
    L = native.lib()
    h = net._handle
    n = C.c_int32()
    native.check(L.cm_debug_conv_count(h, C.byref(n)))
    buf = C.create_string_buffer(512)
    us = C.c_float()
    for i in range(n.value):
        native.check(L.cm_debug_conv_info(h, i, buf, len(buf)))
        f = buf.value.decode().split()
        if f[0] != "conv":
            continue
        label = f[1]
        ntaps, stride, par, Ci, Co, Zo, Yo, Xo, NB, MB0, bz0, by0, bx0, ks, flags = (int(v) for v in f[2:])
        if ntaps not in (27, 8) or (flags & 8):
            continue
        osd = 2 if par else 1
        Z, Y, X = Zo // osd, Yo // osd, Xo // osd
        small = Z * Y * X <= 64                       # K-split layers: the timing includes the combine pass
        if small != bool(os.environ.get("CM_TUNE_SMALL")):
            continue
        fast = bool(flags & 16)
        smalln = bool(flags & 1)
        max_mb = ({1: 5, 2: 2, 4: 1} if fast else {1: 8, 2: 4, 4: 2})[NB]
        native.check(L.cm_debug_time_conv(h, i, 0, 0, 0, 0, B, iters, C.byref(us)))
        base = us.value
        best = (base, MB0, bz0, by0, bx0)
        vox = Z * Y * X
        for bz in range(1, Z + 1):
            for by in range(1, Y + 1):
                for bx in range(1, X + 1):
                    nbox = bz * by * bx
                    MB = (nbox + 31) // 32
                    if MB > max_mb or (smalln and MB & (MB - 1)):
                        continue
                    tiles = -(-Z // bz) * -(-Y // by) * -(-X // bx)
                    if vox / (tiles * 32.0 * MB) < 0.6:
                        continue
                    if L.cm_debug_time_conv(h, i, MB, bz, by, bx, B, iters, C.byref(us)) != 0:
                        continue
                    if us.value < best[0]:
                        best = (us.value, MB, bz, by, bx)
        print("RESULT %s %d %d %d %d %d %d %d %d %d | default MB%d %dx%dx%d %.1f us | best MB%d %dx%dx%d %.1f us" %
              (label, ntaps, stride, par, Ci, Co, Zo, Yo, Xo, NB, MB0, bz0, by0, bx0, base, best[1], best[2], best[3],
               best[4], best[0]), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grids", default="atc")
    ap.add_argument("--channels", default="4")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--child", default=None)
    a = ap.parse_args()
    if a.child:
        grid, ch, nb64, nb128 = a.child.split(",")
        tune_one(grid, int(ch), int(nb64), int(nb128), a.batch, a.iters)
        return
    results = {}
    for grid in a.grids.split(","):
        for ch in a.channels.split(","):
            for nb64, nb128 in ((2, 2), (1, 1)):
                env = dict(os.environ, CM_DIAG="1", CM_NO_TUNED="1", CM_NB64=str(nb64), CM_NB128=str(nb128), CM_LANES="1")
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", f"{grid},{ch},{nb64},{nb128}",
                                      "--batch", str(a.batch), "--iters", str(a.iters)], env=env, capture_output=True, text=True)
                if out.returncode != 0:
                    print(out.stderr[-2000:], file=sys.stderr)
                    raise SystemExit("tuning child failed")
                for line in out.stdout.splitlines():
                    if not line.startswith("RESULT"):
                        continue
                    print(f"# {grid} C={ch} {line[7:]}")
                    head, _, best = line[7:].split("|")
                    f = head.split()
                    key = tuple(int(v) for v in f[1:9])
                    NB = int(f[9])
                    b = best.split()
                    mb = int(b[1][2:])
                    bz, by, bx = (int(v) for v in b[2].split("x"))
                    t = float(b[3])
                    if key not in results or t < results[key][0]:
                        results[key] = (t, NB, mb, bz, by, bx, f[0])
    print("// generated by tools/tune_tiles.py -- {ntaps, stride, par, Ci, Co, Zo, Yo, Xo, NB, MB, bz, by, bx},")
    for key, (t, NB, mb, bz, by, bx, label) in sorted(results.items()):
        print("    {%s, %d, %d, %d, %d, %d},  // %s %.1f us" % (", ".join(str(v) for v in key), NB, mb, bz, by, bx, label, t))


if __name__ == "__main__":
    main()
