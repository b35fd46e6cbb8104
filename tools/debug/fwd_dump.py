"""Dump one full-width forward (ATC grid, B = 8) to a .npy: bit-compare two library builds (CM_LIB_PATH) with np.array_equal."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from crowdmod_ddpm_4d_amd import spec
from crowdmod_ddpm_4d_amd.unet import UNet
B, ch, H, W = 8, 4, int(os.environ.get("TT_H", 12)), int(os.environ.get("TT_W", 36))
net = UNet(input_channels=ch, output_channels=ch, num_res_blocks=1, base_channels=32, base_channels_multiples=(1, 2, 4),
           apply_attention=(False, False, True), max_batch=B)
net.load_state_dict(spec.init_params(net.cfg, 42))
rng = np.random.default_rng(0)
out = net(rng.standard_normal((B, ch, H, W, 3), dtype=np.float32), np.arange(B) * 7 % 1000, rng.standard_normal((B, ch, H, W, 5), dtype=np.float32))
np.save(sys.argv[1], np.asarray(out))
