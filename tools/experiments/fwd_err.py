"""Forward error of the library named by CM_LIB_PATH (and the CM_DIAG switches in the environment) against the reference's own
outputs (tests/golden/fwd.npz): max-abs and rms per grid.  python tools/experiments/fwd_err.py [precision]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from crowdmod_ddpm_4d_amd import spec  # noqa: E402
from crowdmod_ddpm_4d_amd.unet import UNet  # noqa: E402
from helpers import FULL_GRIDS, SEED_W, full_cfg, load, synth_inputs  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
g = load("fwd.npz")
for key in ("atc_c4", "cr120_c3", "atc2x_c3"):
    gname, c = key.split("_c")
    C_ = int(c)
    H, W = FULL_GRIDS[gname]
    cfg = full_cfg(C_)
    net = UNet(cfg.input_channels, cfg.output_channels, cfg.num_res_blocks, cfg.base_channels, cfg.base_channels_multiples,
               cfg.apply_attention, cfg.dropout_rate, cfg.time_multiple, "Past", max_batch=2)
    net.load_state_dict(spec.init_params(cfg, SEED_W))
    net.set_precision(prec)
    past, fut = synth_inputs(2, C_, H, W, 5, 3, f"full/{gname}/c{C_}")
    y = net(fut, g[f"{key}/t"], past)
    e = (y.astype(np.float64) - g[f"{key}/out"])
    print("%-9s %-5s max-abs %.3e  rms %.3e" % (key, prec, np.abs(e).max(), np.sqrt((e ** 2).mean())))
