"""Dump per-block activations of a full-width forward (debug aid): python tools/debug/dump_acts.py out.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from crowdmod_ddpm_4d_amd import spec
from crowdmod_ddpm_4d_amd.unet import UNet
from helpers import full_cfg, synth_inputs, SEED_W
C_, H, W, B = 4, 12, 36, 2
cfg = full_cfg(C_)
net = UNet(cfg.input_channels, cfg.output_channels, cfg.num_res_blocks, cfg.base_channels, cfg.base_channels_multiples,
           cfg.apply_attention, cfg.dropout_rate, cfg.time_multiple, "Past", max_batch=B)
net.load_state_dict(spec.init_params(cfg, SEED_W))
past, fut = synth_inputs(B, C_, H, W, 5, 3, "full/atc/c4")
y = net(fut, np.array([999, 17]), past)
plan = spec.make_plan(cfg)
out = {"y": y}
for name in ["first"] + [b.prefix for b in plan.encoder + plan.bottleneck + plan.decoder] + ["encoder_blocks.2.conv_1", "encoder_blocks.2.conv_2+skip" if False else "encoder_blocks.2"]:
    try:
        out[name] = net.debug_activation(name)[:B]
    except Exception as e:
        print("skip", name, e)
np.savez(sys.argv[1], **out)
