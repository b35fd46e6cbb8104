"""Time explicit tile geometries for selected conv ops (cm_debug_time_conv): quick what-if next to tune_tiles.py.
    python tools/time_tiles.py "<label substring>:MB:bz:by:bx" ..."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from crowdmod_ddpm_4d_amd import native, spec  # noqa: E402
from crowdmod_ddpm_4d_amd.unet import UNet  # noqa: E402

B, ch, H, W = int(os.environ.get("TT_B", 64)), 4, int(os.environ.get("TT_H", 12)), int(os.environ.get("TT_W", 36))
net = UNet(input_channels=ch, output_channels=ch, num_res_blocks=1, base_channels=32, base_channels_multiples=(1, 2, 4),
           apply_attention=(False, False, True), max_batch=B)
net.load_state_dict(spec.init_params(net.cfg, 42))
if os.environ.get("TT_F16"):
    net.set_precision("f16")
rng = np.random.default_rng(0)
net(rng.standard_normal((B, ch, H, W, 3), dtype=np.float32), np.arange(B) * 7 % 1000,
    rng.standard_normal((B, ch, H, W, 5), dtype=np.float32))
L, h = native.lib(), net._handle
n = C.c_int32()
native.check(L.cm_debug_conv_count(h, C.byref(n)))
buf = C.create_string_buffer(512)
us = C.c_float()
for spec_s in sys.argv[1:]:
    key, mb, bz, by, bx = spec_s.split(":")
    for i in range(n.value):
        native.check(L.cm_debug_conv_info(h, i, buf, len(buf)))
        f = buf.value.decode().split()
        if f[0] != "conv" or key not in f[1]:
            continue
        if L.cm_debug_time_conv(h, i, 0, 0, 0, 0, B, 10, C.byref(us)) != 0:
            print(f"{f[1]:44s} not timed: " + L.cm_last_error().decode())
            continue
        base = us.value
        rc = L.cm_debug_time_conv(h, i, int(mb), int(bz), int(by), int(bx), B, 10, C.byref(us))
        print(f"{f[1]:44s} default MB{f[11]} {f[12]}x{f[13]}x{f[14]} {base:7.1f} us | MB{mb} {bz}x{by}x{bx} " +
              (f"{us.value:7.1f} us" if rc == 0 else "rejected: " + L.cm_last_error().decode()))
