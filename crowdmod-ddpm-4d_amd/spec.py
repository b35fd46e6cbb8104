"""Structural description of the reference UNet denoiser.

Everything that depends only on the hyper-parameters lives here: the block
wiring, the `state_dict` tensor names/shapes the reference's checkpoints use
(/root/reference/models/backbones/unet.py:11-122, layers.py:6-53,
embeddings.py:7-31) and a deterministic weight initialiser with PyTorch's
default init *ranges* driven by the repo PRNG.  The oracle, the fixture
generator, the HIP host code and bench.py all build from this one description.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

from . import prng

GN_GROUPS = 8          # layers.py:9,30,41 ; unet.py:119
ATTN_HEADS = 4         # layers.py:10
TIME_TABLE_ROWS = 1000  # embeddings.py:7 (always 1000 rows, regardless of TIMESTEPS)


@dataclass(frozen=True)
class UNetConfig:
    """Hyper-parameters of the reference `UNet` ctor (unet.py:11-25)."""
    input_channels: int = 3
    output_channels: int = 3
    num_res_blocks: int = 1
    base_channels: int = 32
    base_channels_multiples: Tuple[int, ...] = (1, 2, 4)
    apply_attention: Tuple[bool, ...] = (False, False, True, False)
    dropout_rate: float = 0.1
    time_multiple: int = 4
    condition: str = "Past"

    @property
    def time_emb_dims(self) -> int:
        return self.base_channels

    @property
    def time_emb_dims_exp(self) -> int:
        return self.base_channels * self.time_multiple


@dataclass
class Block:
    """One entry of encoder_blocks / bottleneck_blocks / decoder_blocks."""
    kind: str            # "res" | "down" | "up"
    prefix: str          # state_dict prefix, e.g. "encoder_blocks.2"
    cin: int
    cout: int
    attention: bool = False
    skip_channels: int = 0   # decoder res blocks: channels of the popped encoder tensor
    level: int = 0           # resolution level (0 = full)


@dataclass
class UNetPlan:
    cfg: UNetConfig
    encoder: List[Block] = field(default_factory=list)
    bottleneck: List[Block] = field(default_factory=list)
    decoder: List[Block] = field(default_factory=list)
    final_channels: int = 0

    def res_blocks(self) -> List[Block]:
        return [b for b in self.encoder + self.bottleneck + self.decoder if b.kind == "res"]


def make_plan(cfg: UNetConfig) -> UNetPlan:
    """Block wiring, following the loop structure of unet.py:45-115."""
    plan = UNetPlan(cfg)
    base = cfg.base_channels
    nres = len(cfg.base_channels_multiples)
    stack = [base]          # channels of the tensors pushed on `outs` (unet.py:145)
    cin = base
    idx = 0
    for level in range(nres):
        cout = base * cfg.base_channels_multiples[level]
        for _ in range(cfg.num_res_blocks):
            plan.encoder.append(Block("res", f"encoder_blocks.{idx}", cin, cout,
                                      bool(cfg.apply_attention[level]), 0, level))
            idx += 1
            cin = cout
            stack.append(cin)
        if level != nres - 1:
            plan.encoder.append(Block("down", f"encoder_blocks.{idx}", cin, cin, False, 0, level))
            idx += 1
            stack.append(cin)
    plan.bottleneck.append(Block("res", "bottleneck_blocks.0", cin, cin, True, 0, nres - 1))
    plan.bottleneck.append(Block("res", "bottleneck_blocks.1", cin, cin, False, 0, nres - 1))
    idx = 0
    for level in reversed(range(nres)):
        cout = base * cfg.base_channels_multiples[level]
        for _ in range(cfg.num_res_blocks + 1):
            skip = stack.pop()
            plan.decoder.append(Block("res", f"decoder_blocks.{idx}", skip + cin, cout,
                                      bool(cfg.apply_attention[level]), skip, level))
            idx += 1
            cin = cout
        if level != 0:
            plan.decoder.append(Block("up", f"decoder_blocks.{idx}", cin, cin, False, 0, level))
            idx += 1
    plan.final_channels = cin
    return plan


def param_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    """Ordered name -> shape map of the reference `UNet.state_dict()`.

    Conv weights are [Co, Ci, kH, kW, kL] (nn.Conv3d over the reference's
    [B,C,H,W,L] layout); MHA uses the packed in_proj [3E, E].
    """
    plan = make_plan(cfg)
    te, tx = cfg.time_emb_dims, cfg.time_emb_dims_exp
    out: Dict[str, Tuple[int, ...]] = {}
    out["time_embeddings.time_blocks.0.weight"] = (TIME_TABLE_ROWS, te)
    out["time_embeddings.time_blocks.1.weight"] = (tx, te)
    out["time_embeddings.time_blocks.1.bias"] = (tx,)
    out["time_embeddings.time_blocks.3.weight"] = (tx, tx)
    out["time_embeddings.time_blocks.3.bias"] = (tx,)
    out["first.weight"] = (cfg.base_channels, cfg.input_channels, 3, 3, 3)
    out["first.bias"] = (cfg.base_channels,)

    def res(b: Block):
        p = b.prefix
        out[f"{p}.normalize_1.weight"] = (b.cin,)
        out[f"{p}.normalize_1.bias"] = (b.cin,)
        out[f"{p}.conv_1.weight"] = (b.cout, b.cin, 3, 3, 3)
        out[f"{p}.conv_1.bias"] = (b.cout,)
        out[f"{p}.dense_1.weight"] = (b.cout, tx)
        out[f"{p}.dense_1.bias"] = (b.cout,)
        out[f"{p}.normalize_2.weight"] = (b.cout,)
        out[f"{p}.normalize_2.bias"] = (b.cout,)
        out[f"{p}.conv_2.weight"] = (b.cout, b.cout, 3, 3, 3)
        out[f"{p}.conv_2.bias"] = (b.cout,)
        if b.cin != b.cout:
            out[f"{p}.match_input.weight"] = (b.cout, b.cin, 1, 1, 1)
            out[f"{p}.match_input.bias"] = (b.cout,)
        if b.attention:
            out[f"{p}.attention.group_norm.weight"] = (b.cout,)
            out[f"{p}.attention.group_norm.bias"] = (b.cout,)
            out[f"{p}.attention.mhsa.in_proj_weight"] = (3 * b.cout, b.cout)
            out[f"{p}.attention.mhsa.in_proj_bias"] = (3 * b.cout,)
            out[f"{p}.attention.mhsa.out_proj.weight"] = (b.cout, b.cout)
            out[f"{p}.attention.mhsa.out_proj.bias"] = (b.cout,)

    for b in plan.encoder + plan.bottleneck + plan.decoder:
        if b.kind == "res":
            res(b)
        elif b.kind == "down":
            out[f"{b.prefix}.downsample.weight"] = (b.cout, b.cin, 3, 3, 3)
            out[f"{b.prefix}.downsample.bias"] = (b.cout,)
        else:
            out[f"{b.prefix}.upsample.1.weight"] = (b.cout, b.cin, 3, 3, 3)
            out[f"{b.prefix}.upsample.1.bias"] = (b.cout,)
    fc = plan.final_channels
    out["final.0.weight"] = (fc,)
    out["final.0.bias"] = (fc,)
    out["final.2.weight"] = (cfg.output_channels, fc, 3, 3, 3)
    out["final.2.bias"] = (cfg.output_channels,)
    return out


def sinusoid_table(dim: int, rows: int = TIME_TABLE_ROWS) -> np.ndarray:
    """Frozen sin||cos table (embeddings.py:11-20), computed in fp32 like torch does."""
    half = dim // 2
    scale = np.float32(np.log(10000.0) / (half - 1))
    freqs = np.exp(np.arange(half, dtype=np.float32) * -scale).astype(np.float32)
    ang = (np.arange(rows, dtype=np.float32)[:, None] * freqs[None, :]).astype(np.float32)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=-1).astype(np.float32)


def init_params(cfg: UNetConfig, seed: int = 42, gain: float = 1.0,
                perturb_norm: bool = True) -> Dict[str, np.ndarray]:
    """Synthetic fp32 weights with PyTorch's default init *ranges* from the repo PRNG.

    conv / linear weight and bias ~ U(+-1/sqrt(fan_in)) (kaiming_uniform(a=sqrt 5));
    MHA in_proj ~ Xavier-uniform, its biases 0 (torch default).  GroupNorm affine
    defaults are (1, 0); with `perturb_norm` they are jittered so that the parity
    tests can tell gamma from beta and catch a swapped or dropped affine term.
    """
    shapes = param_shapes(cfg)
    params: Dict[str, np.ndarray] = {}
    for name, shp in shapes.items():
        n = int(np.prod(shp))
        if name == "time_embeddings.time_blocks.0.weight":
            params[name] = sinusoid_table(cfg.time_emb_dims)
            continue
        u = prng.uniform_pm1(seed, name, n).reshape(shp)
        leaf = name.rsplit(".", 1)[-1]
        is_norm = ("normalize_" in name) or ("group_norm" in name) or name.startswith("final.0")
        if is_norm:
            if leaf == "weight":
                params[name] = (1.0 + (0.25 * u if perturb_norm else 0.0 * u)).astype(np.float32)
            else:
                params[name] = ((0.1 * u) if perturb_norm else 0.0 * u).astype(np.float32)
            continue
        if leaf == "in_proj_weight":
            bound = np.sqrt(6.0 / (shp[0] + shp[1]))
        elif leaf == "in_proj_bias":
            bound = 0.02 if perturb_norm else 0.0
        elif leaf == "weight":
            fan_in = int(np.prod(shp[1:]))
            bound = gain / np.sqrt(fan_in)
        else:  # bias: fan_in of the matching weight
            wshape = shapes[name[: -len("bias")] + "weight"]
            bound = 1.0 / np.sqrt(int(np.prod(wshape[1:])))
        params[name] = (np.float32(bound) * u).astype(np.float32)
    return params
