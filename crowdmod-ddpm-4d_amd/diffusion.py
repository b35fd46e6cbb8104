"""Host-side mirrors of the reference's diffusion classes.

`ForwardSampler` / `DDPM` expose what /root/reference/models/diffusion/forward.py:9-36
and models/diffusion/ddpm.py:23-38 expose: the six schedule buffers as attributes,
`forward(x0, t) -> (x_t, eps)` and `step(eps_hat, x, t:int) -> (x', sqrt(beta_t),
1-beta_t)`.  Tables come from the native library (cm_schedule_create); the
element-wise updates run on the device.

RNG: the reference draws from torch's global generator.  Here noise is either
passed in explicitly (`noise=`) or drawn from the repo's counter-based stream
(crowdmod-ddpm-4d_amd/prng.py), addressed by (seed, call index): reproducible
and independent of batch sharding.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import native, prng


def get_from_idx(element: np.ndarray, idx) -> np.ndarray:
    """forward.py:4-6: gather + reshape(-1,1,1,1,1)."""
    return np.asarray(element)[np.asarray(idx, dtype=np.int64)].reshape(-1, 1, 1, 1, 1)


class ForwardSampler:
    def __init__(self, timesteps=1000, scale=1, beta_start=1e-4, beta_end=2e-2, *, device: int = 0, seed: int = 0):
        self.timesteps = int(timesteps)
        self.device = int(device)
        self.seed = int(seed)
        self._calls = 0
        h = C.c_void_p()
        native.check(native.lib().cm_schedule_create(self.timesteps, float(scale), float(beta_start), float(beta_end),
                                                     self.device, C.byref(h)))
        self._handle = h
        for i, name in enumerate(native.TABLES):
            buf = np.empty(self.timesteps, dtype=np.float32)
            native.check(native.lib().cm_schedule_table(h, i, buf.ctypes.data, self.timesteps))
            setattr(self, name, buf)

    def to(self, device=None):
        return self

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                native.lib().cm_schedule_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def _noise_like(self, x: np.ndarray, tag: str) -> np.ndarray:
        z = prng.normal(self.seed, f"{tag}/{self._calls}", x.size).reshape(x.shape)
        self._calls += 1
        return z

    def __call__(self, x0, timesteps, noise=None):
        return self.forward(x0, timesteps, noise)

    def forward(self, x0, timesteps, noise=None):
        """forward.py:29-36 (q-sample): returns (x_t, epsilon)."""
        x0 = np.ascontiguousarray(x0, dtype=np.float32)
        B = x0.shape[0]
        eps = self._noise_like(x0, "q") if noise is None else np.ascontiguousarray(noise, dtype=np.float32)
        t = np.ascontiguousarray(np.asarray(timesteps, dtype=np.int64).reshape(-1))
        per = x0.size // B
        dx = native.DeviceBuffer.from_array(x0, self.device)
        de = native.DeviceBuffer.from_array(eps, self.device)
        dt = native.DeviceBuffer.from_array(t, self.device)
        do = native.DeviceBuffer(x0.nbytes, self.device)
        native.check(native.lib().cm_q_sample(self._handle, dx.ptr, dt.ptr, de.ptr, do.ptr, B, per, None))
        native.check(native.lib().cm_device_synchronize(self.device))
        return do.download(x0.shape), eps


class DDPM(ForwardSampler):
    def step(self, predicted_noise, xnoise, timestep: int, noise=None):
        """ddpm.py:25-38: one reverse step; returns (x', sqrt(beta_t), 1 - beta_t)."""
        x = np.ascontiguousarray(xnoise, dtype=np.float32)
        e = np.ascontiguousarray(predicted_noise, dtype=np.float32)
        B = x.shape[0]
        per = x.size // B
        t = int(timestep)
        z = None
        if t > 0:
            z = self._noise_like(x, "z") if noise is None else np.ascontiguousarray(noise, dtype=np.float32)
        dx = native.DeviceBuffer.from_array(x, self.device)
        de = native.DeviceBuffer.from_array(e, self.device)
        dz = native.DeviceBuffer.from_array(z, self.device) if z is not None else None
        if z is None and t > 0:
            raise AssertionError
        native.check(native.lib().cm_ddpm_step(self._handle, de.ptr, dx.ptr, t, dz.ptr if dz else None, 0, 0, B, per,
                                               None))
        native.check(native.lib().cm_device_synchronize(self.device))
        beta_t = self.beta[t]
        return dx.download(x.shape), np.sqrt(beta_t), np.float32(1) - beta_t
