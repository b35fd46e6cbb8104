"""CPU-only tests of the host logic: C-ABI surface, config schemas, checkpoint interop,
schedule tables, multi-process sharding (gloo, world_size 2).  No GPU compute calls."""
import ctypes as C
import os
import re
import socket
import sys
import tempfile
from collections import OrderedDict

import numpy as np
import pytest
import torch

from crowdmod_ddpm_4d_amd import checkpoint, config as cfgmod, distributed, native, prng, spec
from helpers import join_all, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_of_the_header():
    hdr = open(os.path.join(ROOT, "include", "crowdmod_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(cm_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    lib = native.lib()
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    assert names == set(native.SIGNATURES), names ^ set(native.SIGNATURES)
    assert lib.cm_abi_version() == native.ABI_VERSION == 3


def test_error_reporting_without_gpu_is_loud():
    lib = native.lib()
    c = native.cm_unet_config()
    c.in_channels = c.out_channels = 3
    c.num_res_blocks, c.base_channels, c.n_levels = 1, 12, 1  # 12 is not a multiple of 8
    c.channel_mult[0] = 1
    c.time_multiple, c.rows, c.cols, c.past_len, c.future_len, c.max_batch, c.device = 4, 4, 8, 5, 3, 1, 0
    h = C.c_void_p()
    assert lib.cm_model_create(C.byref(c), C.byref(h)) != 0
    assert b"multiple of 8" in lib.cm_last_error()
    with pytest.raises(native.NativeError):
        native.check(lib.cm_model_create(C.byref(c), C.byref(h)))


def test_schedule_tables_host_only_match_reference():
    """cm_schedule_create with device < 0 is pure host code: ForwardSampler.__init__ tables."""
    lib = native.lib()
    g = load("schedule.npz")
    for T, scale in ((1000, 0.5), (50, 0.5), (1000, 1.0)):
        h = C.c_void_p()
        native.check(lib.cm_schedule_create(T, scale, 1e-4, 2e-2, -1, C.byref(h)))
        for i, name in enumerate(native.TABLES):
            buf = np.empty(T, dtype=np.float32)
            native.check(lib.cm_schedule_table(h, i, buf.ctypes.data, T))
            ref = g[f"T{T}_s{scale}/{name}"]
            if name in ("beta", "alpha", "alpha_bar"):
                assert np.array_equal(buf, ref), (T, scale, name)
            else:
                np.testing.assert_allclose(buf, ref, rtol=1.2e-7)
        o = native.cm_sample_opts()
        o.sampler, o.ddim_divider = native.SAMPLER_DDIM, 100
        n = C.c_int32()
        native.check(lib.cm_sample_num_steps(h, C.byref(o), C.byref(n)))
        assert n.value == len(np.arange(0, T - 1, 100))
        o.sampler = native.SAMPLER_DDPM
        native.check(lib.cm_sample_num_steps(h, C.byref(o), C.byref(n)))
        assert n.value == T
        native.check(lib.cm_schedule_destroy(h))


SCHEMA_CURRENT = """
MACROPROPS: {ROWS: 12, COLS: 36}
DATASET: {PAST_LEN: 5, FUTURE_LEN: 3, BATCH_SIZE: 64}
MODEL:
  NSAMPLES: 1280
  NSAMPLES4PLOTS: 20
  DDPM:
    SAMPLER: "DDIM"
    TIMESTEPS: 1000
    SCALE: 0.5
    SIGMA: 0.001
    DDIM_DIVIDER: 2
    GUIDANCE: 'Sparsity'
    LAMBDA_GUIDANCE: 0.004
    UNET: {CONDITION: "Past", NUM_RES_BLOCKS: 1, BASE_CH: 32, BASE_CH_MULT: [1, 2, 4],
           APPLY_ATTENTION: [False, False, True, False], DROPOUT_RATE: 0.1, TIME_EMB_MULT: 4,
           TRAIN: {EPOCHS: 200, SOLVER: {LR: 0.00005}}}
"""
SCHEMA_4TEST = """
MACROPROPS: {ROWS: 12, COLS: 36}
DATASET: {PAST_LEN: 5, FUTURE_LEN: 3, BATCH_SIZE: 8}
MODEL:
  CONDITION: "Past"
  NUM_RES_BLOCKS: 1
  BASE_CH: 32
  BASE_CH_MULT: [1, 2, 4]
  APPLY_ATTENTION: [False, False, True, False]
  DROPOUT_RATE: 0.1
  TIME_EMB_MULT: 4
  DDPM: {SAMPLER: "DDPM", TIMESTEPS: 200, SCALE: 0.5, GUIDANCE: 'sparsity', TRAIN: {EPOCHS: 3}}
"""
SCHEMA_FLAT = """
MACROPROPS: {ROWS: 12, COLS: 36}
DATASET: {PAST_LEN: 5, FUTURE_LEN: 3, BATCH_SIZE: 64}
MODEL: {CONDITION: "Past", NUM_RES_BLOCKS: 2, BASE_CH: 16, BASE_CH_MULT: [1, 2], APPLY_ATTENTION: [False, True],
        DROPOUT_RATE: 0.05, TIME_EMB_MULT: 4}
DIFFUSION: {SAMPLER: "DDPM", TIMESTEPS: 1000, SCALE: 0.5, DDIM_DIVIDER: 90, NSAMPLES: 640, NSAMPLES4PLOTS: 10,
            GUIDANCE: 'mass_preservation'}
TRAIN: {EPOCHS: 250, SOLVER: {LR: 0.00005}}
"""


def _cfg(text):
    with tempfile.NamedTemporaryFile("w", suffix=".yml", delete=False) as f:
        f.write(text)
    try:
        return cfgmod.getYamlConfig(f.name)
    finally:
        os.unlink(f.name)


def test_config_three_schema_generations():
    r = cfgmod.resolve(_cfg(SCHEMA_CURRENT))
    assert (r.rows, r.cols, r.timesteps, r.sampler, r.guidance, r.lambda_guidance) == (12, 36, 1000, "DDIM", "Sparsity", 0.004)
    assert r.base_ch_mult == (1, 2, 4) and r.apply_attention == (False, False, True, False) and r.train.EPOCHS == 200
    r = cfgmod.resolve(_cfg(SCHEMA_4TEST))
    assert (r.timesteps, r.base_ch, r.batch_size, r.guidance) == (200, 32, 8, "sparsity") and r.train.EPOCHS == 3
    r = cfgmod.resolve(_cfg(SCHEMA_FLAT))
    assert (r.num_res_blocks, r.base_ch, r.base_ch_mult, r.nsamples, r.ddim_divider) == (2, 16, (1, 2), 640, 90)
    assert r.train.EPOCHS == 250
    shipped = cfgmod.resolve(cfgmod.getYamlConfig(os.path.join(ROOT, "config", "ATC.yml")))
    assert (shipped.rows, shipped.cols, shipped.base_ch, shipped.timesteps) == (12, 36, 32, 1000)
    hermes = cfgmod.resolve(cfgmod.getYamlConfig(os.path.join(ROOT, "config", "HERMES-CR-120.yml")))
    assert (hermes.rows, hermes.cols) == (28, 24)


def test_attr_dict_and_missing_file():
    cfg = _cfg(SCHEMA_CURRENT)
    assert cfg.MODEL.DDPM.UNET.BASE_CH == 32 and cfg["MODEL"]["DDPM"]["SCALE"] == 0.5
    with pytest.raises(AttributeError):
        _ = cfg.MODEL.DDPM.NOPE
    with pytest.raises(FileNotFoundError):
        cfgmod.getYamlConfig("/nonexistent.yml")


def test_checkpoint_interop_both_directions(tmp_path):
    cfg = spec.UNetConfig(3, 3, 1, 8, (1, 2, 4), (False, False, True, False))
    P = spec.init_params(cfg, 1)
    sd = OrderedDict((k, torch.from_numpy(v.copy())) for k, v in P.items())
    a = str(tmp_path / "a.pth")
    torch.save({"opt": {"state": {}, "param_groups": [{"lr": 5e-5, "betas": (0.5, 0.999)}]}, "model": sd}, a)
    m = checkpoint.load_model_state(a)
    assert list(m) == list(P) and all(np.array_equal(m[k], P[k]) for k in P)
    t = torch.arange(24, dtype=torch.float32).reshape(4, 6).t()          # non-contiguous view
    b = str(tmp_path / "b.pth")
    torch.save({"model": {"x": t}}, b)
    assert np.array_equal(checkpoint.load_model_state(b)["x"], t.numpy())
    c = str(tmp_path / "c.pth")
    checkpoint.save_checkpoint(P, c, opt_state={"state": {}, "param_groups": [{"lr": 5e-5, "betas": (0.5, 0.999)}]})
    o = torch.load(c, map_location="cpu", weights_only=True)
    assert list(o["model"]) == list(P) and all(np.array_equal(o["model"][k].numpy(), P[k]) for k in P)
    assert o["opt"]["param_groups"][0]["betas"] == (0.5, 0.999)


def test_checkpoint_rejects_code_execution(tmp_path):
    import pickle
    import zipfile

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))

    p = str(tmp_path / "evil.pth")
    with zipfile.ZipFile(p, "w") as zf:
        zf.writestr("archive/data.pkl", pickle.dumps({"model": Evil()}, protocol=2))
        zf.writestr("archive/version", "3\n")
    with pytest.raises(pickle.UnpicklingError):
        checkpoint.load(p)


def _tensor_pickle(size, stride, offset, numel=6):
    """A torch-zip checkpoint whose single tensor claims the given view geometry over a 6-float storage."""
    import struct

    def i4(v):
        if -(1 << 31) <= v < (1 << 31):
            return b"J" + struct.pack("<i", v)
        raw = v.to_bytes((v.bit_length() + 8) // 8, "little", signed=True)     # LONG1
        return b"\x8a" + bytes([len(raw)]) + raw

    def tup(vals):
        return b"(" + b"".join(i4(v) for v in vals) + b"t"

    pk = (b"\x80\x02}" + b"X\x05\x00\x00\x00model" + b"}" + b"X\x01\x00\x00\x00w"
          + b"ctorch._utils\n_rebuild_tensor_v2\n" + b"("
          + b"(" + b"X\x07\x00\x00\x00storage" + b"ctorch\nFloatStorage\n" + b"X\x01\x00\x00\x000"
          + b"X\x03\x00\x00\x00cpu" + i4(numel) + b"t" + b"Q"
          + i4(offset) + tup(size) + tup(stride) + b"\x89" + b"ccollections\nOrderedDict\n)R" + b"t" + b"R"
          + b"s" + b"s" + b".")
    return pk


@pytest.mark.parametrize("size,stride,offset,ok", [
    ((2, 3), (3, 1), 0, True),           # the honest view
    ((2, 3), (3, 1), 1, False),          # offset pushes the last element out of the storage
    ((2, 3), (1000, 1), 0, False),       # stride reaches far outside
    ((2, 3), (-1, 1), 3, False),         # negative stride
    ((-2, 3), (3, 1), 0, False),         # negative size
    ((2, 3), (3, 1), -1, False),         # negative offset
    ((2, 3), (0, 1), 0, True),           # broadcast view with no more elements than the storage: legal
    ((1 << 20, 3), (0, 1), 0, False),    # broadcast view LARGER than its storage: would be materialised -> rejected
    ((1 << 40,), (0,), 0, False),        # ... before a terabyte allocation, not after
    ((0, 3), (3, 1), 100, True),         # empty tensor: nothing is read
])
def test_checkpoint_rejects_out_of_bounds_tensor_views(tmp_path, size, stride, offset, ok):
    """A crafted .pth must not make the loader read outside the storage bytes of its zip entry
    (torch's weights_only path validates the same geometry)."""
    import pickle
    import zipfile
    p = str(tmp_path / "view.pth")
    with zipfile.ZipFile(p, "w") as zf:
        zf.writestr("archive/data.pkl", _tensor_pickle(size, stride, offset))
        zf.writestr("archive/data/0", np.arange(6, dtype=np.float32).tobytes())
        zf.writestr("archive/version", "3\n")
    if ok:
        w = checkpoint.load(p)["model"]["w"]
        assert w.shape == tuple(size)
        if w.size and size == (2, 3) and stride == (3, 1):
            assert np.array_equal(w, np.arange(6, dtype=np.float32).reshape(2, 3))
    else:
        with pytest.raises(pickle.UnpicklingError):
            checkpoint.load(p)


def test_host_only_handle_lists_the_state_dict_without_a_gpu():
    """device < 0: the state_dict plan (names, shapes, set / get) works on the CPU; finalize refuses."""
    lib = native.lib()
    cfg = spec.UNetConfig(3, 3, 1, 32, (1, 2, 4), (False, False, True, False), 0.1, 4, "Past")
    c = native.cm_unet_config()
    c.in_channels = c.out_channels = 3
    c.num_res_blocks, c.base_channels, c.n_levels = 1, 32, 3
    for i, (mlt, att) in enumerate(zip((1, 2, 4), (0, 0, 1))):
        c.channel_mult[i], c.apply_attention[i] = mlt, att
    c.time_multiple, c.rows, c.cols, c.past_len, c.future_len, c.max_batch, c.device = 4, 12, 36, 5, 3, 2, -1
    h = C.c_void_p()
    native.check(lib.cm_model_create(C.byref(c), C.byref(h)))
    n = C.c_int32()
    native.check(lib.cm_model_num_params(h, C.byref(n)))
    shapes = spec.param_shapes(cfg)
    assert n.value == len(shapes) == 169
    names = []
    for i in range(n.value):
        name, shp, nd = C.c_char_p(), (C.c_int64 * 5)(), C.c_int32()
        native.check(lib.cm_model_param_info(h, i, C.byref(name), shp, C.byref(nd)))
        names.append(name.value.decode())
        assert tuple(shp[: nd.value]) == tuple(shapes[names[-1]])
    assert names == list(shapes)
    w = np.arange(32, dtype=np.float32)
    native.check(lib.cm_model_set_param(h, b"first.bias", w.ctypes.data, w.size))
    back = np.empty_like(w)
    native.check(lib.cm_model_get_param(h, b"first.bias", back.ctypes.data, back.size))
    assert np.array_equal(w, back)
    assert lib.cm_model_set_param(h, b"first.bias", w.ctypes.data, 31) != 0 and b"size mismatch" in lib.cm_last_error()
    assert lib.cm_model_finalize(h) != 0 and b"host-only" in lib.cm_last_error()
    native.check(lib.cm_model_destroy(h))


def test_unet_host_mirror_state_dict_contract():
    from crowdmod_ddpm_4d_amd.unet import UNet
    net = UNet(3, 3, 1, 8, (1, 2, 4), (False, False, True, False), 0.1, 4, "Past")
    sd = net.state_dict()
    assert list(sd) == list(spec.param_shapes(net.cfg)) and len(sd) == 169
    with pytest.raises(RuntimeError, match="missing keys"):
        net.load_state_dict({k: v for k, v in sd.items() if k != "first.bias"})
    bad = dict(sd)
    bad["first.bias"] = np.zeros(9, np.float32)
    with pytest.raises(RuntimeError, match="size mismatch"):
        net.load_state_dict(bad)
    with pytest.raises(NotImplementedError):
        UNet(3, 3, 1, 8, (1, 2, 4), (False, False, True, False), condition="None")


def test_prng_is_order_and_shard_independent():
    a = prng.normal_per_sample(7, "z/x", np.arange(8), 100, step=3)
    b = prng.normal_per_sample(7, "z/x", np.arange(4, 8), 100, step=3)
    assert np.array_equal(a[4:], b)
    assert not np.array_equal(a[0], prng.normal_per_sample(7, "z/x", [0], 100, step=4)[0])
    z = prng.normal(1, "moments", 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01


def test_shard_ranges_partition_the_batch():
    for gb in (1, 7, 64, 512, 1280):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = distributed.shard_range(gb, r, world)
                cover += list(range(lo, hi))
            assert cover == list(range(gb))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, gb, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = 12
    past = prng.normal(3, "past", gb * per).reshape(gb, per)

    def generate(p, n, base):
        # stands in for the on-device loop: depends on the chain's past and its GLOBAL index only
        z = prng.normal_per_sample(9, "z", np.arange(base, base + n), per, step=0)
        return (np.tanh(p) + 0.5 * z).astype(np.float32)

    full = distributed.sample_sharded(generate, past, gb, rank, world)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("gb", [8, 7])
def test_gloo_two_ranks_sharded_sampling_equals_single_process(tmp_path, gb):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, gb, str(tmp_path)), nprocs=world, join=True)
    per = 12
    past = prng.normal(3, "past", gb * per).reshape(gb, per)
    want = (np.tanh(past) + 0.5 * prng.normal_per_sample(9, "z", np.arange(gb), per, step=0)).astype(np.float32)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"r{r}.npy"))
        assert np.array_equal(got, want), r


def test_plateau_scheduler_matches_torch():
    """ReduceLROnPlateau(mode='min', factor, patience, min_lr) as the reference builds it (ddpm.py:58-63)."""
    import torch
    from crowdmod_ddpm_4d_amd.ddpm_model import ReduceLROnPlateau
    p = [torch.nn.Parameter(torch.zeros(1))]
    o = torch.optim.SGD(p, lr=5e-5)
    ts = torch.optim.lr_scheduler.ReduceLROnPlateau(o, mode="min", factor=0.5, patience=3, min_lr=1e-6)
    mine = ReduceLROnPlateau(5e-5, 0.5, 3, 1e-6)
    rng = np.random.default_rng(3)
    for i in range(120):
        v = 1.0 / (1 + 0.05 * i) + (0.2 if i > 20 else 0.0) + 0.01 * rng.random()
        ts.step(v)
        assert abs(o.param_groups[0]["lr"] - mine.step(v)) < 1e-18, i


def test_checkpoint_opt_entry_loads_into_torch_adam(tmp_path):
    """The "opt" entry written next to "model" (utils/utils.py:140-147) must be a torch.optim.Adam
    state_dict the reference can resume from: torch.load(weights_only=True) + load_state_dict."""
    import torch
    from crowdmod_ddpm_4d_amd import checkpoint
    shapes = {"a.weight": (4, 3), "b.weight": (5,), "c.bias": (2, 2, 2)}
    model = {k: np.arange(int(np.prod(s)), dtype=np.float32).reshape(s) for k, s in shapes.items()}
    names = list(shapes)
    state = {i: {"step": np.float32(7), "exp_avg": np.full(shapes[n], 0.5, np.float32),
                 "exp_avg_sq": np.full(shapes[n], 0.25, np.float32)} for i, n in enumerate(names) if i > 0}
    group = {"lr": 2.5e-5, "betas": (0.5, 0.999), "eps": 1e-8, "weight_decay": 0.003, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "decoupled_weight_decay": False, "params": [0, 1, 2]}
    path = str(tmp_path / "ck.pth")
    checkpoint.save_checkpoint(model, path, opt_state={"state": state, "param_groups": [group]})
    ck = torch.load(path, map_location="cpu", weights_only=True)
    params = [torch.nn.Parameter(torch.from_numpy(model[n].copy())) for n in names]
    params[0].requires_grad_(False)
    opt = torch.optim.Adam(params, lr=1.0, betas=(0.9, 0.9), weight_decay=0.0)
    opt.load_state_dict(ck["opt"])
    assert opt.param_groups[0]["lr"] == 2.5e-5 and tuple(opt.param_groups[0]["betas"]) == (0.5, 0.999)
    assert float(opt.state[params[1]]["step"]) == 7.0
    assert torch.equal(opt.state[params[2]]["exp_avg_sq"], torch.full((2, 2, 2), 0.25))
    assert params[0] not in opt.state
    # and our own reader returns the same structure
    back = checkpoint.load(path)
    assert set(back["opt"]["state"].keys()) == {1, 2}
    assert np.array_equal(back["model"]["c.bias"], model["c.bias"])


def _dp_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from crowdmod_ddpm_4d_amd import distributed as cdist
    import crowdmod_ddpm_4d_amd.native as native
    cdist.init_process_group("gloo")
    # stand-in for the device buffer: the averager's host path only needs memcpy d2h / h2d
    buf = np.full(1000, float(rank + 1), dtype=np.float32)

    class FakeLib:
        def cm_memcpy_d2h(self, dev, dst, src, nbytes):
            C.memmove(dst, src, nbytes)
            return 0

        def cm_memcpy_h2d(self, dev, dst, src, nbytes):
            C.memmove(dst, src, nbytes)
            return 0

    native.lib = lambda: FakeLib()
    native.check = lambda rc: None

    class Net:
        device = 0

        def flat_grads(self):
            return buf.ctypes.data, buf.size

    cdist.GradAverager()(Net())
    q.put((rank, float(buf[0]), float(buf[-1])))
    dist.destroy_process_group()


def test_gradient_averaging_two_ranks_gloo():
    """Data-parallel training: mean of the flat gradient buffers over 2 ranks (gloo on CPU; the GPU path
    reduces the same buffer in place with RCCL)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    try:
        res = sorted(q.get(timeout=120) for _ in ps)
    finally:
        codes = join_all(ps, 60)
    assert codes == [0, 0]
    assert res == [(0, 1.5, 1.5), (1, 1.5, 1.5)]


def _dp_control_worker(rank, world, port, q):
    """Epoch-level control flow of DDPM_model.train under data parallelism, with the device step stubbed out:
    rank-dependent epoch losses must still give identical scheduler / stop decisions on every rank."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from crowdmod_ddpm_4d_amd import distributed as cdist, ddpm_model as dm
    cdist.init_process_group("gloo")
    cfg = cfgmod.getYamlConfig(os.path.join(ROOT, "config", "ATC.yml"))
    cfg.MODEL.DDPM.UNET.TRAIN.EPOCHS = 40
    model = dm.DDPM_model(cfg, "DDPM-UNet", 3, device=-1)   # device -1: host-only schedule handle
    model.set_data_parallel(rank, world)
    lrs, tdraws = [], []

    class Net:   # stands in for the native denoiser: records what the control flow tells it
        def set_lr(self, lr): lrs.append(lr)
        def sync_trained(self): pass

    model.denoiser = Net()

    def one_epoch(fs, loader, epoch, **kw):
        if not hasattr(model, "_plateau"):
            s = model._solver()
            model._lr = s["lr"]
            model._plateau = dm.ReduceLROnPlateau(s["lr"], s["factor"], 2, s["min_lr"])
        rng = np.random.default_rng([model.seed + epoch, model.dp_rank])
        tdraws.append(int(rng.integers(0, 1000)))
        # rank 0 plateaus alone from epoch 5 on; rank 1 keeps improving until epoch 20; NaN only on rank 1 at the end
        base = 1.0 / epoch if (rank == 1 and epoch < 20) or epoch < 5 else 0.2
        return float("nan") if (rank == 1 and epoch >= 30) else base

    model._train_one_epoch = one_epoch
    hist = model.train([], None, save=False, loss_sync=cdist.mean_over_ranks)
    q.put((rank, len(hist), lrs, model._lr, tdraws[:4]))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_epoch_decisions_are_identical_on_every_rank():
    """ReduceLROnPlateau, the 3-NaN stop and the learning rate must not diverge between replicas
    (decisions on the rank-averaged loss), while the timestep draws must differ between ranks."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_dp_control_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    try:
        res = sorted(q.get(timeout=180) for _ in ps)
    finally:
        codes = join_all(ps, 60)
    assert codes == [0, 0]
    (r0, n0, lrs0, lr0, t0), (r1, n1, lrs1, lr1, t1) = res
    assert n0 == n1 == 32          # NaN from epoch 30 on rank 1 only -> both stop after 3 NaN epochs
    assert lrs0 == lrs1 and len(lrs0) >= 1 and lr0 == lr1 < 5e-5
    assert t0 != t1                # per-rank timestep streams


def test_host_side_under_address_sanitizer():
    """`make asan`: the host half of the library (plan builder, weight / index packers, tile planner and its
    coordinate tables, schedule, error paths) under ASan + UBSan as a self-test binary."""
    import subprocess
    csrc = os.path.join(ROOT, "crowdmod-ddpm-4d_amd", "csrc")
    exe = os.path.join(csrc, "asan", "cm_host_selftest")
    r = subprocess.run(["make", "-C", csrc, "-j4", "asan"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "selftest ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_self_launch_builds_a_torchrun_child_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts N ranks as a CHILD process tree (never an exec of a
    process that touched the GPU) with the rendezvous on 127.0.0.1 and forwards its own flags."""
    import subprocess
    import types
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5"])
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "5"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_energy_metric_matches_the_reference_compute_energy():
    """models/guidance.py:10-42 restated in numpy (crowdmod-ddpm-4d_amd/metrics.py) against the reference's outputs
    (tests/golden/energy.npz, generated by importing the reference's module)."""
    from crowdmod_ddpm_4d_amd import metrics
    g = load("energy.npz")
    B, C_, H, W, L = 6, 3, 12, 36, 3
    x = prng.normal(7, "energy/x", B * C_ * H * W * L).reshape(B, C_, H, W, L).astype(np.float32)
    x[:, 0] = np.maximum(x[:, 0], 0.0)
    np.testing.assert_allclose(metrics.compute_energy(x, 1, 1), g["e11"], rtol=2e-6)
    np.testing.assert_allclose(metrics.compute_energy(x), g["e_default"], rtol=2e-6)
    t = metrics.energy_tables(x[:4] * 0.9, x[:4], 2)
    assert t["ENERGY"].shape == (4, 2) and t["MIN-ENERGY"].shape == (2, 2)
    np.testing.assert_allclose(t["ENERGY"][:, 0], g["e11"][:4], rtol=2e-6)
    assert np.allclose(t["MIN-ENERGY"][0], t["ENERGY"][:2].min(axis=0))


def test_ssim_restatement_against_brute_force_and_its_identities():
    """SSIM (Wang et al. 2004, skimage defaults: 7x7 uniform window, sample covariance, border cropped) -- the library the
    reference calls is not installed, so the restatement is checked against a direct per-window evaluation of the formula
    and the metric's identities (SSIM(x, x) = 1, symmetry, monotone under noise)."""
    from crowdmod_ddpm_4d_amd import metrics
    rng = np.random.default_rng(3)
    x = rng.normal(size=(12, 36))
    y = x + 0.3 * rng.normal(size=x.shape)
    R = float(x.max() - x.min())
    win, pad = 7, 3
    C1, C2 = (0.01 * R) ** 2, (0.03 * R) ** 2
    vals = []
    for i in range(pad, x.shape[0] - pad):
        for j in range(pad, x.shape[1] - pad):
            a = x[i - pad:i + pad + 1, j - pad:j + pad + 1].ravel()
            b = y[i - pad:i + pad + 1, j - pad:j + pad + 1].ravel()
            ua, ub = a.mean(), b.mean()
            va, vb = a.var(ddof=1), b.var(ddof=1)
            vab = ((a - ua) * (b - ub)).sum() / (a.size - 1)
            vals.append(((2 * ua * ub + C1) * (2 * vab + C2)) / ((ua * ua + ub * ub + C1) * (va + vb + C2)))
    brute = float(np.mean(vals))
    got = metrics._ssim2d(x, y, R)
    assert abs(got - brute) <= 1e-10
    assert abs(metrics._ssim2d(x, x, R) - 1.0) <= 1e-12
    assert abs(metrics._ssim2d(y, x, R) - got) <= 1e-12
    assert metrics._ssim2d(x, x + 1.0 * rng.normal(size=x.shape), R) < got < 1.0
    N, F = 4, 3
    gt = rng.normal(size=(N, 3, 12, 36, F)).astype(np.float32)
    pred = gt + 0.2 * rng.normal(size=gt.shape).astype(np.float32)
    t = metrics.ssim_tables(pred, gt, [6.0, 6.0, 6.0], 2)
    assert t["SSIM"].shape == (N, 3) and t["SSIM_OVER_TIME"].shape == (N, 9) and t["MAX_SSIM"].shape == (2, 3)
    assert np.allclose(t["SSIM"][:, 1], t["SSIM_OVER_TIME"][:, 1::3].mean(axis=1))
    assert np.allclose(t["MAX_SSIM_OVER_TIME"][1], t["SSIM_OVER_TIME"][2:4].max(axis=0))
    with pytest.raises(ValueError):
        metrics._ssim2d(x[:5], y[:5], R)


def test_motion_feature_metrics_match_the_reference_extractor():
    """MF_MSE / MF_BHATT (missing item of the round-2 review): the vectorised motion-feature histograms against the vectors
    and tables the reference's own MotionFeatureExtractor / MetricsGenerator produced on the same sequences
    (tests/golden/motion_feat.npz, generated by importing the reference) -- ATC setting and a ragged one (f 2, k 5, gamma 2),
    with motionless cells, constant cells and angles of exactly +-pi in the data.  Counts are integers: exact."""
    from crowdmod_ddpm_4d_amd import metrics
    g = load("motion_feat.npz")
    for tag in ("atc", "ragged"):
        f, k, gamma = g[f"{tag}/fkg"]
        mag, ang = metrics._mf_polar(g["pred"])
        N = mag.shape[0]
        np.testing.assert_array_equal(mag.reshape(N, mag.shape[1], -1), g[f"{tag}/mag"])
        np.testing.assert_array_equal(ang.reshape(N, ang.shape[1], -1), g[f"{tag}/ang"])
        p2, p1 = metrics.motion_feature_vectors(g["pred"], int(f), int(k), float(gamma))
        g2, g1 = metrics.motion_feature_vectors(g["gt"], int(f), int(k), float(gamma))
        np.testing.assert_array_equal(p2, g[f"{tag}/p2"])
        np.testing.assert_array_equal(g2, g[f"{tag}/g2"])
        np.testing.assert_allclose(p1, g[f"{tag}/p1"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(g1, g[f"{tag}/g1"], rtol=1e-12, atol=1e-15)
        t = metrics.motion_feature_tables(g["pred"], g["gt"], int(f), int(k), float(gamma))
        for name in ("MF_MSE", "MF_BHATT_DIST", "MF_BHATT_COEF"):
            assert t[name].shape == (N, 2)
            np.testing.assert_allclose(t[name], g[f"{tag}/{name}"], rtol=1e-10, atol=1e-15)
    # identical sequences: coefficient sum(P) = S / (S + 1) < 1, distance -log of it; disjoint ones clip at epsilon
    same = metrics.motion_feature_tables(g["gt"], g["gt"], 1, 4, 0.5)
    assert np.all(same["MF_MSE"] == 0) and np.all(same["MF_BHATT_COEF"] <= 1.0) and np.all(same["MF_BHATT_COEF"] > 0.99)
    d, c = metrics.bhattacharyya(np.array([[1.0, 0.0]]), np.array([[0.0, 1.0]]))
    assert c[0] == 1e-2 and abs(d[0] + np.log(1e-2)) < 1e-15
    only = metrics.motion_feature_tables(g["pred"], g["gt"], 1, 4, 0.5, mse_metric=False, bhatt_metrics=True)
    assert only["MF_MSE"] is None and only["MF_BHATT_DIST"] is not None


def test_bench_secondary_object_has_one_record_per_other_baseline_config(monkeypatch):
    """bench.py's `secondary` object (round-3 verdict item 2): one record per BASELINE config besides the headline, each with
    ms_per_step / value / dtype / workload / roofline.frac; a failing secondary measurement becomes an `error` record and
    never takes the headline line down; the whole object serialises."""
    import argparse
    import json
    import bench
    calls = []

    def fake_sampling(cfg_path, grid, channels, batch, dtype, steps, warmup, repeats, lanes, device=0):
        calls.append((cfg_path, grid, channels, batch, dtype, steps))
        if batch == 2:
            raise RuntimeError("boom")
        return {"ms_per_step": 2.0, "value": 500.0, "unit": "denoise-steps/s", "dtype": dtype, "steps": steps, "repeats": repeats,
                "batch": batch, "channels": channels, "grid": list(grid or (12, 36)), "roofline": {"frac": 0.3}}

    def fake_train(a):
        assert a.batch == 128 and a.steps >= 50
        return {"config": {"workload": "train"}, "ms_per_step": 12.0, "value": 83.0, "dtype": "f32",
                "roofline": {"frac": 0.4, "bound": "mfma", "achieved": 60.0, "peak": 157.3, "unit": "TFLOP/s"}}

    monkeypatch.setattr(bench, "measure_sampling", fake_sampling)
    monkeypatch.setattr(bench, "measure_train", fake_train)
    sec = bench.measure_secondary(argparse.Namespace(lanes=2))
    assert set(sec) == {"configs[3]", "configs[4]", "configs[4]_f32", "configs[0]", "configs[2]", "configs[1]_f32r", "configs[1]_f32x"}
    assert ("config/ATC.yml", None, 4, 64, "f32x", 50) in calls and sec["configs[1]_f32x"]["dtype"] == "f32x"
    assert ("config/HERMES-CR-120.yml", None, 3, 64, "f32", 50) in calls
    assert ("config/ATC.yml", None, 4, 64, "f32r", 50) in calls and sec["configs[1]_f32r"]["dtype"] == "f32r"
    assert ("config/ATC_synthetic.yml", (24, 72), 3, 32, "f16", 50) in calls
    for k in ("configs[3]", "configs[4]", "configs[4]_f32", "configs[2]"):
        r = sec[k]
        assert r["ms_per_step"] > 0 and r["value"] > 0 and r["dtype"] in ("f32", "f16") and r["workload"] and 0 < r["roofline"]["frac"] <= 1
    assert "boom" in sec["configs[0]"]["error"]
    json.dumps(sec)


def test_h2_split_simulation_matches_the_six_term_form():
    """tools/sim_six_term.py, the numpy model behind DESIGN section 4: fp32 products from f16 TWO-way splits (three cross terms,
    weights packed as w * 2^13) are as close to fp64 as the six-term bf16 form on a Winograd layer's operand distribution, and
    three bf16 terms (the opt-in relaxed plan) are several times further away."""
    import importlib.util
    import os
    spec_ = importlib.util.spec_from_file_location("sim_six_term", os.path.join(os.path.dirname(__file__), "..", "tools", "sim_six_term.py"))
    sim = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(sim)
    out = sim.h2_vs_six(K=288, N=512, seed=3)
    six, h2, h2s, three, f32 = (out[k][0] for k in ("six-term bf16", "h2 (f16 x 2, 3 terms)", "h2, weights x 2^13", "three-term bf16", "fp32 chain"))
    assert h2 <= 1.15 * six and h2s <= 1.15 * six, (h2, h2s, six)
    assert six <= 4.0 * f32 and three >= 2.5 * six, (six, f32, three)
