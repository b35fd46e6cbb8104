"""Debug aid (not a test): per-block difference between the HIP training forward and the CPU oracle on one of the
geometries of test_gpu_ref_geometries.py.   python tests/debug_train_fwd.py atc_medium [eval]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from crowdmod_ddpm_4d_amd import prng, spec  # noqa: E402
from crowdmod_ddpm_4d_amd.unet import UNet  # noqa: E402
from helpers import SEED_W, synth_inputs  # noqa: E402
from oracle import unet_torch as ot  # noqa: E402
from test_gpu_ref_geometries import GEOMS, _cfg  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "atc_medium"
evalmode = len(sys.argv) > 2
B = 2
H, W, P_, F_, base, att, C = GEOMS[name]
cfg = _cfg(name)
plan = spec.make_plan(cfg)
params = spec.init_params(cfg, SEED_W)
past, fut = synth_inputs(B, C, H, W, P_, F_, f"geom/{name}")
t = np.array([3, 777])
masks = {}
for blk in plan.res_blocks():
    u = prng.uniform_pm1(11, f"dropgeom/{blk.prefix}", B * blk.cout).reshape(B, blk.cout)
    masks[blk.prefix] = ((u * 0.5 + 0.5) >= 0.1).astype(np.float32) / np.float32(0.9)
Pt = ot.to_torch(params)
dm = None if evalmode else {k: torch.tensor(v) for k, v in masks.items()}
acts = {}
with torch.no_grad():
    tt = torch.tensor(t, dtype=torch.long)
    temb = ot.time_embedding(tt, Pt)
    x = torch.cat([torch.tensor(past), torch.tensor(fut)], dim=4)
    h = F.conv3d(x, Pt["first.weight"], Pt["first.bias"], padding=1)
    acts["first"] = h
    outs = [h]

    def run(blk, h):
        if blk.kind == "res":
            return ot._res_block(h, temb, Pt, blk.prefix, None if dm is None else dm.get(blk.prefix))
        if blk.kind == "down":
            return F.conv3d(h, Pt[blk.prefix + ".downsample.weight"], Pt[blk.prefix + ".downsample.bias"], stride=2, padding=1)
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        return F.conv3d(h, Pt[blk.prefix + ".upsample.1.weight"], Pt[blk.prefix + ".upsample.1.bias"], padding=1)

    for blk in plan.encoder:
        h = run(blk, h)
        outs.append(h)
        acts[blk.prefix] = h
    for blk in plan.bottleneck:
        h = run(blk, h)
        acts[blk.prefix] = h
    for blk in plan.decoder:
        if blk.kind == "res":
            h = torch.cat([h, outs.pop()], dim=1)
        h = run(blk, h)
        acts[blk.prefix] = h

net = UNet(input_channels=C, output_channels=C, num_res_blocks=1, base_channels=base, base_channels_multiples=(1, 2, 4),
           apply_attention=att, dropout_rate=0.1, time_multiple=4, condition="Past", max_batch=B)
net.load_state_dict(params)
if evalmode:
    net(fut, t, past)
else:
    net.ensure(H, W, P_, F_, B)
    net.train()
    net.forward_train(fut, t, past, drop_masks=masks)
for k, v in acts.items():
    try:
        got = net.debug_activation(k)[:B]
    except Exception as e:  # noqa: BLE001
        print(f"{k:28s} not available: {e}")
        continue
    ref = v.numpy()
    print(f"{k:28s} shape {tuple(ref.shape)} max|ref| {np.abs(ref).max():9.4f} max|diff| {np.abs(got - ref).max():.3e}")
