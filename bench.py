#!/usr/bin/env python3
"""Headline benchmark: denoise-steps/sec of the DDPM-UNet reverse loop.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): config/ATC.yml sampling, batch 64 per GPU,
T = 1000 schedule, tensors [B,4,12,36,(5 past + 3 future)] -- synthetic inputs,
random-init weights (torch-default init ranges from the repo PRNG).  A "step" is
one pass of the hot path over the batch: 1 UNet forward + 1 sampler update
(/root/reference/models/diffusion/ddpm.py:214-221).  K steps are the first K
visited timesteps of the 1000-step loop, run on the device behind ONE C-ABI call.

N > 1: launched by torch.distributed.run, one rank per GPU; the batch of
independent chains is sharded (64 per GPU, weak scaling), the only collective is
one RCCL all_gather of x at the end of the region.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the 3x3x3 MFMA
conv, timed with HIP events on its launch stream) and, at N=1, `cpu_baseline`
(this repo's torch-functional CPU restatement of the reference path, i.e. the
same ATen CPU kernels the reference dispatches to, on the host cores).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak


def load_cfg(channels: int):
    from crowdmod_ddpm_4d_amd import config as cfgmod
    cfg = cfgmod.getYamlConfig(os.path.join(ROOT, "config", "ATC.yml"))
    return cfg, cfgmod.resolve(cfg, "DDPM-UNet")


def hbm_traffic():
    """HBM bytes per launch of the dominant kernel class, from the rocprofv3 PMC passes of
    tools/profile_round.sh (FETCH_SIZE x2-corrected + WRITE_SIZE; see profiles/hbm_traffic.json).
    bench.py cannot collect PMC counters on itself, so the committed measurement is quoted."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as fo:
        return json.load(fo).get("hbm_bytes_per_launch")


def cpu_baseline(res, channels: int, batch: int, budget_s: float):
    """Time the CPU port (oracle/unet_torch.py) on a bounded sample of the same workload."""
    import torch
    from crowdmod_ddpm_4d_amd import prng, spec
    from oracle import unet_torch as ot
    cfg = spec.UNetConfig(channels, channels, res.num_res_blocks, res.base_ch, res.base_ch_mult, res.apply_attention,
                          res.dropout_rate, res.time_emb_mult, "Past")
    P = ot.to_torch(spec.init_params(cfg, 42, perturb_norm=False))
    plan = spec.make_plan(cfg)
    sched = ot.schedule(res.timesteps, res.scale)
    shape_f = (batch, channels, res.rows, res.cols, res.future_len)
    shape_p = (batch, channels, res.rows, res.cols, res.past_len)
    past = torch.from_numpy(prng.normal(7, "bench/past", int(np.prod(shape_p))).reshape(shape_p))
    x = torch.from_numpy(prng.normal(7, "bench/xT", int(np.prod(shape_f))).reshape(shape_f))
    z = torch.from_numpy(prng.normal(7, "bench/z", int(np.prod(shape_f))).reshape(shape_f))
    T = res.timesteps
    ot.generate_ddpm(P, plan, sched, past, x, lambda t: z, T, t_list=[T - 1])  # warm-up step
    n, t0 = 0, time.perf_counter()
    while True:
        x = ot.generate_ddpm(P, plan, sched, past, x, lambda t: z, T, t_list=[T - 2 - n])
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 50:
            break
    return {"value": n / el, "unit": "denoise-steps/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"{n} steps of the B={batch} ATC loop after 1 warm-up step (torch {torch.__version__} CPU, "
                      f"oracle/unet_torch.py)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="chains per GPU")
    ap.add_argument("--channels", type=int, default=4)
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU baseline work (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"WORLD_SIZE={world} but --gpus={a.gpus}")

    import torch
    from crowdmod_ddpm_4d_amd import native, prng
    from crowdmod_ddpm_4d_amd.ddpm_model import DDPM_model
    from crowdmod_ddpm_4d_amd.diffusion import DDPM

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # rehearsal switches (one-GPU box): CM_BENCH_BACKEND=gloo CM_BENCH_SHARE_GPU=1 run N ranks on cuda:0 to
    # exercise the multi-rank control flow; the driver's N-GPU runs use neither
    backend = os.environ.get("CM_BENCH_BACKEND", "nccl")
    if os.environ.get("CM_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    cfg, res = load_cfg(a.channels)
    Cn, B = a.channels, a.batch
    model = DDPM_model(cfg, "DDPM-UNet", Cn, device=local_rank, seed=42)
    model.denoiser.max_batch = B
    sampler = DDPM(timesteps=res.timesteps, scale=res.scale, device=local_rank)
    dev = torch.device("cuda", local_rank)
    gid0 = rank * B  # global index of this shard's first chain
    shape_p = (B, Cn, res.rows, res.cols, res.past_len)
    per_p = int(np.prod(shape_p[1:]))
    past = torch.from_numpy(prng.normal_per_sample(7, "bench/past", np.arange(gid0, gid0 + B), per_p).reshape(shape_p)).to(dev)

    def run(nsteps):
        x, _ = model._generate_ddpm(past, sampler, B, sample_id_base=gid0, first_steps=nsteps)
        return x

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if a.warmup > 0:
        run(a.warmup)
    sync()
    t0 = time.perf_counter()
    x = run(a.steps)
    if dist is not None:
        xg = x if backend == "nccl" else x.cpu()
        gathered = [torch.empty_like(xg) for _ in range(world)]
        dist.all_gather(gathered, xg)  # the trivial gather of the sharded result (RCCL over xGMI)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- roofline of the dominant kernel: a profiled re-run of the same K steps ------
    roofline = None
    if not a.no_profile and rank == 0:
        L = native.lib()
        h = model.denoiser._handle
        native.check(L.cm_profile_enable(h, 1))
        run(a.steps)
        torch.cuda.synchronize()
        ms = (C.c_float * 8)()
        cnt = (C.c_int64 * 8)()
        native.check(L.cm_profile_read(h, ms, cnt))
        if os.environ.get("CM_BENCH_REPORT"):
            buf = C.create_string_buffer(1 << 16)
            native.check(L.cm_profile_report(h, buf, len(buf)))
            with open(os.environ["CM_BENCH_REPORT"], "w") as fo:
                fo.write(buf.value.decode())
        native.check(L.cm_profile_enable(h, 0))
        fl = C.c_double()
        by = C.c_double()
        native.check(L.cm_model_cost(h, B, C.byref(fl), C.byref(by)))
        conv3_flops = model.denoiser.conv3_flops(B) * a.steps
        conv_s = ms[0] / 1e3
        ach = conv3_flops / conv_s / 1e12 if conv_s > 0 else 0.0
        roofline = {
            "kernel": "conv_mfma_kernel<*,*,27|8|0> -- all 3x3x3 conv launches (implicit GEMM, v_mfma_f32_32x32x2_f32)",
            "bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": hbm_traffic(),
            "launches": int(cnt[0]), "avg_launch_us": ms[0] * 1e3 / max(1, cnt[0]),
            "measured_with": "second run of the same K steps, same launch configuration, with HIP events around every "
                             "launch on its launch stream (events inside the timed run would add ~1.5 us per "
                             "launch to `value`); rocprofv3 --kernel-trace --stats of this command: "
                             "profiles/round1_kernel_stats.csv",
            "algorithmic_gflop_per_launch": conv3_flops / max(1, cnt[0]) / 1e9,
            "class_ms_per_step": {k: ms[i] / a.steps for i, k in enumerate(
                ["conv3x3x3", "conv1x1x1_gemm", "groupnorm_stats", "attention_core", "elementwise"]) },
            "step_algorithmic_gflop": fl.value / 1e9, "step_algorithmic_gbytes": by.value / 1e9,
        }
    if dist is not None:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and a.cpu_budget > 0:
        cpu = cpu_baseline(res, Cn, B, a.cpu_budget)

    if rank == 0:
        out = {
            "metric": "denoise-steps/sec (UNet fwd + sampler update) at ATC [B,4,T,H,W]",
            "value": world * a.steps / elapsed,
            "unit": "denoise-steps/s (each over a batch of %d chains)" % B,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "sample_steps_per_s": world * B * a.steps / elapsed,
            "config": {"workload": "config/ATC.yml sampling (BASELINE configs[1]): DDPM p_sample_loop, T=1000, "
                                   "batch %d per GPU, UNet base 32 mult [1,2,4]" % B,
                       "channels": Cn, "grid": [res.rows, res.cols], "past_len": res.past_len,
                       "future_len": res.future_len, "global_batch": world * B, "parallelism": "batch-shard x%d" % world},
        }
        if roofline:
            out["roofline"] = roofline
        if cpu:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
