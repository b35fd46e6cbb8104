import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from crowdmod_ddpm_4d_amd import native, spec
from crowdmod_ddpm_4d_amd.unet import UNet
B, ch, H, W = 4, 4, 24, 72
net = UNet(input_channels=ch, output_channels=ch, num_res_blocks=1, base_channels=32, base_channels_multiples=(1, 2, 4),
           apply_attention=(False, False, True), max_batch=B)
net.load_state_dict(spec.init_params(net.cfg, 42))
rng = np.random.default_rng(0)
net(rng.standard_normal((B, ch, H, W, 3), dtype=np.float32), np.arange(B) * 7 % 1000, rng.standard_normal((B, ch, H, W, 5), dtype=np.float32))
L, h = native.lib(), net._handle
n = C.c_int32(); native.check(L.cm_debug_conv_count(h, C.byref(n)))
buf = C.create_string_buffer(512)
for i in range(n.value):
    native.check(L.cm_debug_conv_info(h, i, buf, len(buf)))
    s = buf.value.decode()
    if "proj" in s or "attention" in s: print(i, s)
