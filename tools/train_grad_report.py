"""Debug aid: per-tensor gradient norms of one training step of the narrow model next to the
reference's (tests/golden/train.npz).  Run on the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T  # noqa: E402

g = T.load("train.npz")
m, sampler, past, fut, eps, masks = T._narrow_train_setup()
net = m.denoiser
net.ensure(T.NARROW["H"], T.NARROW["W"], T.NARROW["P"], T.NARROW["F"], 4)
net.train_init(lr=5e-5, betas=(0.5, 0.999), weight_decay=0.003)
loss = net.train_step(sampler._handle, fut, past, g["t"], eps, drop_masks=masks, apply_update=False)
print("loss", loss, float(g["loss"]))
for key in g.files:
    if key.startswith("gnorm/"):
        name = key[6:]
        ref = float(g[key])
        got = float(np.sqrt((net.grad(name).astype(np.float64) ** 2).sum()))
        flag = "" if abs(got - ref) <= 1e-3 * ref + 2e-7 else "  <<<<"
        print(f"{name:60s} {got:12.5e} {ref:12.5e}{flag}")
for key in g.files:
    if key.startswith("grad/"):
        name = key[5:]
        ref = g[key]
        got = net.grad(name)
        print(f"full {name:50s} maxerr {np.abs(got - ref).max():.3e}  max|ref| {np.abs(ref).max():.3e}")
