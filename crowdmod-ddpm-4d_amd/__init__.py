"""MI355X-native DDPM-UNet denoiser + sampler for crowd macroproperty sequences.

Drop-in for the hot path of marcemq/crowdmod-ddpm-4D: `UNet.forward`,
`ForwardSampler` / `DDPM.step` and `DDPM_model._generate_ddpm/_generate_ddim`,
running as hand-written HIP (gfx950) behind a C-ABI (`include/crowdmod_hip.h`).

The directory name carries a hyphen (it is fixed by the project layout), so it
is imported through the root-level shim module `crowdmod_ddpm_4d_amd`.
"""
from . import prng, spec  # noqa: F401

__all__ = ["prng", "spec"]
from . import config  # noqa: F401,E402


def __getattr__(name):
    # native-backed modules are imported lazily so that `import crowdmod_ddpm_4d_amd`
    # works on a box without the built library (e.g. for prng / spec / config only)
    if name in ("native", "unet", "diffusion", "ddpm_model"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    if name in ("UNet",):
        from .unet import UNet
        return UNet
    if name in ("DDPM", "ForwardSampler"):
        from . import diffusion
        return getattr(diffusion, name)
    if name == "DDPM_model":
        from .ddpm_model import DDPM_model
        return DDPM_model
    raise AttributeError(name)
