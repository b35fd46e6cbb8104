"""Config loading with the reference's surface: `getYamlConfig(cfg.yml, datalist.yml)`
returns an attribute-dict (/root/reference/utils/myparser.py:5-33).

The reference's configs come in three schema generations;
`resolve()` extracts the keys the DDPM-UNet path needs from any of them.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Tuple

import yaml


class AttrDict(dict):
    """Minimal EasyDict: nested dicts become attribute-accessible."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(AttrDict(i) if isinstance(i, dict) else i for i in v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def update(self, d=None, **kw):
        for k, v in dict(d or {}, **kw).items():
            self[k] = v


class YamlParser(AttrDict):
    def __init__(self, cfg_dict=None, config_file=None):
        super().__init__(cfg_dict or {})
        if config_file is not None:
            self.merge_from_file(config_file)

    def merge_from_file(self, config_file):
        if not os.path.isfile(config_file):
            raise FileNotFoundError(config_file)
        with open(config_file, "r") as fo:
            self.update(yaml.safe_load(fo.read()) or {})

    def merge_from_dict(self, config_dict):
        self.update(config_dict)


def get_config(config_file=None):
    return YamlParser(config_file=config_file)


def getYamlConfig(config_yml_file, configList_yml_file=None):
    cfg = get_config()
    cfg.merge_from_file(config_yml_file)
    if configList_yml_file is not None:
        loaded = yaml.safe_load(open(configList_yml_file).read())
        if isinstance(loaded, dict):
            cfg.update(loaded)
    return cfg


@dataclass
class Resolved:
    """The hot path's view of a config, independent of schema generation."""
    rows: int
    cols: int
    past_len: int
    future_len: int
    batch_size: int
    timesteps: int
    scale: float
    sampler: str
    sigma: float
    ddim_divider: int
    guidance: str
    lambda_guidance: float
    num_res_blocks: int
    base_ch: int
    base_ch_mult: Tuple[int, ...]
    apply_attention: Tuple[bool, ...]
    dropout_rate: float
    time_emb_mult: int
    condition: str
    nsamples: int
    nsamples4plots: int
    train: Optional[AttrDict] = None


def _first(*vals, default=None):
    for v in vals:
        if v is not None:
            return v
    return default


def resolve(cfg, arch: str = "DDPM-UNet") -> Resolved:
    """Accept MODEL.DDPM.UNET.* (current), MODEL.DDPM.* + MODEL.* (4test) and the flat
    MODEL.* / DIFFUSION.* / TRAIN.* generation."""
    model = cfg.get("MODEL", {})
    gen_key, back_key = arch.upper().split("-")
    gen = model.get(gen_key, {}) or {}
    back = gen.get(back_key, {}) or {}
    diff = cfg.get("DIFFUSION", {}) or {}

    def bk(key, default=None):
        return _first(back.get(key), gen.get(key), model.get(key), default=default)

    def df(key, default=None):
        return _first(gen.get(key), diff.get(key), model.get(key), default=default)

    train = _first(back.get("TRAIN"), gen.get("TRAIN"), cfg.get("TRAIN"))
    mult = tuple(int(v) for v in bk("BASE_CH_MULT", (1, 2, 4)))
    attn = tuple(bool(v) for v in bk("APPLY_ATTENTION", (False, False, True, False)))
    return Resolved(
        rows=int(cfg.MACROPROPS.ROWS), cols=int(cfg.MACROPROPS.COLS),
        past_len=int(cfg.DATASET.PAST_LEN), future_len=int(cfg.DATASET.FUTURE_LEN),
        batch_size=int(cfg.DATASET.get("BATCH_SIZE", 64)),
        timesteps=int(df("TIMESTEPS", 1000)), scale=float(df("SCALE", 0.5)),
        sampler=str(df("SAMPLER", "DDPM")), sigma=float(df("SIGMA", 0.0)),
        ddim_divider=int(df("DDIM_DIVIDER", 1)), guidance=str(df("GUIDANCE", "None")),
        lambda_guidance=float(df("LAMBDA_GUIDANCE", 0.0)),
        num_res_blocks=int(bk("NUM_RES_BLOCKS", 1)), base_ch=int(bk("BASE_CH", 32)),
        base_ch_mult=mult, apply_attention=attn, dropout_rate=float(bk("DROPOUT_RATE", 0.1)),
        time_emb_mult=int(bk("TIME_EMB_MULT", 4)), condition=str(bk("CONDITION", "Past")),
        nsamples=int(_first(model.get("NSAMPLES"), diff.get("NSAMPLES"), default=1280)),
        nsamples4plots=int(_first(model.get("NSAMPLES4PLOTS"), diff.get("NSAMPLES4PLOTS"), default=20)),
        train=train,
    )
