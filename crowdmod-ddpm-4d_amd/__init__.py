"""MI355X-native DDPM-UNet denoiser + sampler for crowd macroproperty sequences.

Drop-in for the hot path of marcemq/crowdmod-ddpm-4D: `UNet.forward`,
`ForwardSampler` / `DDPM.step` and `DDPM_model._generate_ddpm/_generate_ddim`,
running as hand-written HIP (gfx950) behind a C-ABI (`include/crowdmod_hip.h`).

The directory name carries a hyphen (it is fixed by the project layout), so it
is imported through the root-level shim module `crowdmod_ddpm_4d_amd`.
"""
from . import prng, spec  # noqa: F401

__all__ = ["prng", "spec"]
