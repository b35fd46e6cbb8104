"""Host-side mirror of the reference denoiser interface.

`UNet` has the constructor signature and call convention of
/root/reference/models/backbones/unet.py:7-25,124 --
`denoiser(future[B,C,H,W,F], t[B] int64, past[B,C,H,W,P]) -> [B,C,H,W,F]` -- and the
`nn.Module` methods the reference's drivers touch (`eval/train/to/state_dict/
load_state_dict/parameters`, models/diffusion/ddpm.py:50,118,209,288).  All
arithmetic happens in libcrowdmod_hip.so; this class only owns the weights on
the host and the native handle.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Optional

import numpy as np

from . import native, spec


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


class UNet:
    def __init__(self, input_channels=4, output_channels=4, num_res_blocks=2, base_channels=128,
                 base_channels_multiples=(1, 2, 4, 8), apply_attention=(False, False, True, False, False),
                 dropout_rate=0.1, time_multiple=4, condition="Past", *, device: int = 0, max_batch: int = 64,
                 seed: Optional[int] = 42):
        if condition != "Past":
            raise NotImplementedError("only condition='Past' (the configuration every reference config uses)")
        self.cfg = spec.UNetConfig(int(input_channels), int(output_channels), int(num_res_blocks), int(base_channels),
                                   tuple(int(v) for v in base_channels_multiples),
                                   tuple(bool(v) for v in apply_attention), float(dropout_rate), int(time_multiple),
                                   condition)
        self.input_channels = self.cfg.input_channels
        self.condition = condition
        self.device = int(device)
        self.max_batch = int(max_batch)
        self._native_max_batch = 0
        self.training = False
        self._shapes = spec.param_shapes(self.cfg)
        # random init with torch-default ranges (the reference's nn.Module ctor does the same)
        self._params: Dict[str, np.ndarray] = spec.init_params(self.cfg, seed if seed is not None else 0,
                                                               perturb_norm=False)
        self._handle = None
        self._geom = None
        self.precision = "f32"

    def set_precision(self, precision: str):
        """'f32' (default: fp32 arithmetic) or 'f16': f16 matrix-core operands with fp32 accumulation in the stride-1
        3x3x3 layers, the upsample convs and -- on grids whose attention runs the generic chain -- the attention core and
        in-projection of the inference plan: the counterpart of the reference's torch.amp.autocast (ddpm.py:116-120)."""
        # 'f32r' (relaxed fp32): fp32 tensors and accumulation, but the six-term layers keep only their three leading cross terms
        # (~16 mantissa bits per product; include/crowdmod_hip.h, CM_PRECISION_F32R).  Inside the 1e-4 bound against the
        # reference, not inside the default plan's 2e-6.  Inference only.
        # 'f32x' (strict fp32, round 3's arithmetic): the default plan without the f16 two-way-split form -- exact three-way bf16
        # splits, six cross terms, in every split layer (CM_PRECISION_F32X): same measured error, 16 % slower.
        if precision not in ("f32", "f16", "f32r", "f32x"):
            raise ValueError(f"precision {precision!r}: 'f32', 'f32x', 'f32r' or 'f16'")
        if precision != self.precision:
            self._release()
            self.precision = precision
        return self

    # -- nn.Module surface ---------------------------------------------------------
    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    def to(self, device=None):
        if isinstance(device, int):
            if device != self.device:
                self._release()
                self.device = device
        return self

    def parameters(self) -> Iterable[np.ndarray]:
        return [v for k, v in self._params.items() if k != "time_embeddings.time_blocks.0.weight"]

    def state_dict(self) -> Dict[str, np.ndarray]:
        return {k: v.copy() for k, v in self._params.items()}

    def load_state_dict(self, state: Dict[str, object], strict: bool = True):
        got = {}
        for k, v in state.items():
            if _is_torch(v):
                v = v.detach().cpu().numpy()
            got[k] = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
        missing = [k for k in self._shapes if k not in got]
        unexpected = [k for k in got if k not in self._shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for UNet: missing keys {missing}, "
                               f"unexpected keys {unexpected}")
        for k, shp in self._shapes.items():
            if k in got:
                if tuple(got[k].shape) != tuple(shp):
                    raise RuntimeError(f"size mismatch for {k}: got {tuple(got[k].shape)}, expected {tuple(shp)}")
                self._params[k] = got[k]
        self._carry_opt = None
        self._release(keep_training=False)
        return self

    # -- native handle -------------------------------------------------------------
    def _release(self, keep_training: bool = True):
        """Destroy the native handle.  A handle that has trained holds the only copy of the master weights
        and the Adam moments: they are pulled back to the host first and re-installed by the next ensure()
        (keep_training=False -- load_state_dict -- discards them on purpose)."""
        if self._handle is not None:
            if getattr(self, "_train_ready", False) and keep_training:
                self.sync_trained()
                hp = self._train_hparams
                self._carry_opt = (self.optimizer_state_dict(hp["lr"], hp["betas"], hp["eps"], hp["weight_decay"]), hp)
            native.lib().cm_model_destroy(self._handle)
            self._handle = None
            self._geom = None
            self._train_ready = False

    def __del__(self):
        try:
            self._release(keep_training=False)
        except Exception:
            pass

    def ensure(self, rows: int, cols: int, past_len: int, future_len: int, batch: int):
        """Create (or re-create) the native model for this tensor geometry."""
        geom = (rows, cols, past_len, future_len)
        # compare against the capacity the NATIVE handle was created with, not the Python attribute: a caller that
        # raised `max_batch` after the handle existed must get a new handle, not a rejected launch
        if self._handle is not None and self._geom == geom and max(batch, self.max_batch) <= self._native_max_batch:
            return self._handle
        self._release()
        self.max_batch = max(self.max_batch, batch)
        L = native.lib()
        c = native.cm_unet_config()
        c.in_channels, c.out_channels = self.cfg.input_channels, self.cfg.output_channels
        c.num_res_blocks, c.base_channels = self.cfg.num_res_blocks, self.cfg.base_channels
        c.n_levels = len(self.cfg.base_channels_multiples)
        for i, v in enumerate(self.cfg.base_channels_multiples):
            c.channel_mult[i] = v
            c.apply_attention[i] = int(self.cfg.apply_attention[i])
        c.time_multiple = self.cfg.time_multiple
        c.rows, c.cols, c.past_len, c.future_len = rows, cols, past_len, future_len
        c.max_batch, c.device = self.max_batch, self.device
        h = C.c_void_p()
        native.check(L.cm_model_create(C.byref(c), C.byref(h)))
        try:
            if self.precision == "f16":
                native.check(L.cm_model_set_precision(h, native.PRECISION_F16))
            elif self.precision == "f32r":
                native.check(L.cm_model_set_precision(h, native.PRECISION_F32R))
            elif self.precision == "f32x":
                native.check(L.cm_model_set_precision(h, native.PRECISION_F32X))
            for name, arr in self._params.items():
                arr = np.ascontiguousarray(arr, dtype=np.float32)
                native.check(L.cm_model_set_param(h, name.encode(), arr.ctypes.data, arr.size))
            native.check(L.cm_model_finalize(h))
        except Exception:
            L.cm_model_destroy(h)
            raise
        self._handle, self._geom, self._native_max_batch = h, geom, self.max_batch
        carry = getattr(self, "_carry_opt", None)
        if carry is not None:
            # the handle this one replaces was training: continue from its optimizer state
            opt, hp = carry
            self._carry_opt = None
            self.train_init(lr=hp["lr"], betas=hp["betas"], eps=hp["eps"], weight_decay=hp["weight_decay"])
            self.load_optimizer_state_dict(opt)
        return h

    # -- forward -------------------------------------------------------------------
    def __call__(self, future, t, past=None):
        return self.forward(future, t, past)

    def forward(self, future, t, past=None):
        """UNet.forward (unet.py:124-167), eval mode.  numpy in -> numpy out (host
        staging); torch CUDA tensors in -> torch CUDA tensor out (device pointers)."""
        if past is None:
            raise ValueError("condition='Past' needs the past frames")
        if self.training:
            return self.forward_train(future, t, past)
        L = native.lib()
        B, Cc, H, W, F = (int(v) for v in future.shape)
        P = int(past.shape[4])
        if Cc != self.cfg.input_channels or tuple(past.shape[:4]) != (B, Cc, H, W):
            raise ValueError(f"shape mismatch: future {tuple(future.shape)}, past {tuple(past.shape)}")
        h = self.ensure(H, W, P, F, B)
        if _is_torch(future):
            import torch
            if not future.is_cuda:
                raise ValueError("torch inputs must live on the GPU; pass numpy arrays for host staging")
            fut = future.contiguous().float()
            pst = past.contiguous().float()
            tt = t.to(device=future.device, dtype=torch.long).contiguous()
            out = torch.empty_like(fut)
            # the library runs on its own stream: order it after the producers of the
            # inputs and hand back a finished result
            torch.cuda.current_stream(future.device).synchronize()
            native.check(L.cm_unet_forward(h, fut.data_ptr(), tt.data_ptr(), pst.data_ptr(), out.data_ptr(), B, None))
            native.check(L.cm_device_synchronize(self.device))
            return out
        fut = np.ascontiguousarray(future, dtype=np.float32)
        pst = np.ascontiguousarray(past, dtype=np.float32)
        tt = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.int64).reshape(-1), (B,)))
        out = np.empty_like(fut)
        native.check(L.cm_unet_forward_host(h, fut.ctypes.data, tt.ctypes.data, pst.ctypes.data, out.ctypes.data, B))
        return out

    def dropout_layout(self):
        """[(prefix, offset, Cout)] of the per-block slices of the training-mode mask row."""
        plan = spec.make_plan(self.cfg)
        out, off = [], 0
        for b in plan.res_blocks():
            out.append((b.prefix, off, b.cout))
            off += b.cout
        return out, off

    def forward_train(self, future, t, past, drop_masks=None, seed: int = 0, sample_id_base: int = 0):
        """Training-mode forward (Dropout3d active, layers.py:42,71).  `drop_masks` maps a
        ResnetBlock prefix to its [B, Cout] keep-mask/(1-p); None draws the masks on the device.
        numpy in -> numpy out."""
        L = native.lib()
        fut = np.ascontiguousarray(future, dtype=np.float32)
        pst = np.ascontiguousarray(past, dtype=np.float32)
        B, Cc, H, W, F = fut.shape
        P = pst.shape[4]
        h = self.ensure(H, W, P, F, B)
        tt = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.int64).reshape(-1), (B,)))
        dev = self.device
        dfut, dpst, dt = (native.DeviceBuffer.from_array(a, dev) for a in (fut, pst, tt))
        dout = native.DeviceBuffer(fut.nbytes, dev)
        dmask = None
        if drop_masks is not None:
            layout, width = self.dropout_layout()
            row = np.ones((B, width), dtype=np.float32)
            for prefix, off, cout in layout:
                row[:, off:off + cout] = np.asarray(drop_masks[prefix], dtype=np.float32)
            dmask = native.DeviceBuffer.from_array(row, dev)
        native.check(L.cm_unet_forward_train(h, dfut.ptr, dt.ptr, dpst.ptr, dmask.ptr if dmask else None,
                                             float(self.cfg.dropout_rate), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                             int(sample_id_base), dout.ptr, B, None))
        native.check(L.cm_device_synchronize(dev))
        return dout.download(fut.shape)

    # -- training step (ddpm.py:111-121,142-144) -------------------------------------
    def train_init(self, lr=5e-5, betas=(0.5, 0.999), eps=1e-8, weight_decay=0.003):
        """Device-side optimizer state: torch.optim.Adam(lr, betas, weight_decay) of ddpm.py:53-56."""
        if self._handle is None:
            raise RuntimeError("call ensure() (or run a forward) before train_init()")
        native.check(native.lib().cm_train_init(self._handle, float(lr), float(betas[0]), float(betas[1]), float(eps),
                                                float(weight_decay), float(self.cfg.dropout_rate)))
        self._train_ready = True
        self._train_hparams = {"lr": float(lr), "betas": (float(betas[0]), float(betas[1])), "eps": float(eps),
                               "weight_decay": float(weight_decay)}
        if getattr(self, "_sample_base", 0):
            native.check(native.lib().cm_train_set_sample_base(self._handle, int(self._sample_base)))

    def set_lr(self, lr: float):
        native.check(native.lib().cm_train_set_lr(self._handle, float(lr)))
        self._train_hparams["lr"] = float(lr)

    def set_sample_base(self, sample_id_base: int):
        """Data-parallel training: global index of this rank's sample 0, so that the device-drawn eps and
        Dropout3d masks differ between ranks (and do not depend on how the job is sharded)."""
        self._sample_base = int(sample_id_base)
        if getattr(self, "_train_ready", False):
            native.check(native.lib().cm_train_set_sample_base(self._handle, self._sample_base))

    def _mask_rows(self, B, drop_masks):
        layout, width = self.dropout_layout()
        row = np.ones((B, width), dtype=np.float32)
        for prefix, off, cout in layout:
            row[:, off:off + cout] = np.asarray(drop_masks[prefix], dtype=np.float32)
        return row

    def train_step(self, schedule_handle, future, past, t, eps, drop_masks=None, seed: int = 0,
                   apply_update: bool = True) -> float:
        """q-sample + train-mode forward + MSE + backward (+ Adam) on the device; returns the loss.
        Inputs are numpy arrays (host staging) or native.DeviceBuffer objects already in HBM."""
        L = native.lib()
        dev = self.device

        def dbuf(a, dtype):
            if isinstance(a, native.DeviceBuffer):
                return a
            return native.DeviceBuffer.from_array(np.ascontiguousarray(a, dtype=dtype), dev)

        B = int(t.nbytes // 8) if isinstance(t, native.DeviceBuffer) else int(np.asarray(t).size)
        if not getattr(self, "_train_ready", False):
            raise RuntimeError("train_init() has not been called")
        dfut, dpst = dbuf(future, np.float32), dbuf(past, np.float32)
        deps = dbuf(eps, np.float32) if eps is not None else None  # None: drawn on the device
        dt = dbuf(t, np.int64)
        dmask = None
        if drop_masks is not None:
            dmask = dbuf(self._mask_rows(B, drop_masks) if isinstance(drop_masks, dict) else drop_masks, np.float32)
        loss = C.c_float()
        native.check(L.cm_train_step(self._handle, schedule_handle, dfut.ptr, dpst.ptr, dt.ptr,
                                     deps.ptr if deps else None,
                                     dmask.ptr if dmask else None, int(seed) & 0xFFFFFFFFFFFFFFFF, C.byref(loss), B,
                                     1 if apply_update else 0, None))
        return float(loss.value)

    def train_step_xt(self, xt, past, t, target, drop_masks=None, seed: int = 0, apply_update: bool = True) -> float:
        """The training step with the caller's noised input and regression target (flow matching):
        loss = mse(UNet_train(xt, t, past), target); backward; Adam."""
        L = native.lib()
        dev = self.device

        def dbuf(a, dtype):
            if isinstance(a, native.DeviceBuffer):
                return a
            return native.DeviceBuffer.from_array(np.ascontiguousarray(a, dtype=dtype), dev)

        B = int(t.nbytes // 8) if isinstance(t, native.DeviceBuffer) else int(np.asarray(t).size)
        if not getattr(self, "_train_ready", False):
            raise RuntimeError("train_init() has not been called")
        dxt, dpst, dtg, dt = dbuf(xt, np.float32), dbuf(past, np.float32), dbuf(target, np.float32), dbuf(t, np.int64)
        dmask = None
        if drop_masks is not None:
            dmask = dbuf(self._mask_rows(B, drop_masks) if isinstance(drop_masks, dict) else drop_masks, np.float32)
        loss = C.c_float()
        native.check(L.cm_train_step_xt(self._handle, dxt.ptr, dpst.ptr, dt.ptr, dtg.ptr, dmask.ptr if dmask else None,
                                        int(seed) & 0xFFFFFFFFFFFFFFFF, C.byref(loss), B, 1 if apply_update else 0, None))
        return float(loss.value)

    def grad(self, name: str) -> np.ndarray:
        """Gradient of a state_dict tensor after the last train_step (reference layout)."""
        shape = self._params[name].shape
        out = np.empty(shape, dtype=np.float32)
        native.check(native.lib().cm_train_get_grad(self._handle, name.encode(), out.ctypes.data, out.size))
        return out

    def flat_grads(self):
        """(device pointer, numel) of the flat fp32 gradient buffer in state_dict order (data-parallel
        all-reduce between train_step(apply_update=False) and apply_update())."""
        ptr, n = C.c_void_p(), C.c_int64()
        native.check(native.lib().cm_train_flat_grads(self._handle, C.byref(ptr), C.byref(n)))
        return ptr.value, int(n.value)

    def apply_update(self):
        native.check(native.lib().cm_train_apply(self._handle, None))

    def trainable_names(self):
        """Names in model.parameters() order; index 0 (the frozen sinusoid table) carries no Adam state."""
        return list(self._params.keys())

    def optimizer_state_dict(self, lr: float, betas, eps: float, weight_decay: float):
        """torch.optim.Adam.state_dict() of the reference optimizer (ddpm.py:53-56): what
        save_checkpoint stores under "opt" (utils/utils.py:140-147)."""
        L = native.lib()
        step = C.c_int32()
        native.check(L.cm_train_opt_step(self._handle, C.byref(step), 0))
        names = self.trainable_names()
        state = {}
        if step.value > 0:
            for i, name in enumerate(names):
                if i == 0:
                    continue  # requires_grad=False: the optimizer never created state for it
                shp = self._params[name].shape
                m1, m2 = np.empty(shp, np.float32), np.empty(shp, np.float32)
                native.check(L.cm_train_get_opt_state(self._handle, name.encode(), 0, m1.ctypes.data, m1.size))
                native.check(L.cm_train_get_opt_state(self._handle, name.encode(), 1, m2.ctypes.data, m2.size))
                state[i] = {"step": np.float32(step.value), "exp_avg": m1, "exp_avg_sq": m2}
        group = {"lr": float(lr), "betas": (float(betas[0]), float(betas[1])), "eps": float(eps),
                 "weight_decay": float(weight_decay), "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, opt: dict):
        """Inverse of optimizer_state_dict (resume from a checkpoint's "opt" entry)."""
        L = native.lib()
        names = self.trainable_names()
        step = 0
        for i, st in opt.get("state", {}).items():
            name = names[int(i)]
            for which, key in ((0, "exp_avg"), (1, "exp_avg_sq")):
                arr = np.ascontiguousarray(np.asarray(st[key]), dtype=np.float32)
                native.check(L.cm_train_set_opt_state(self._handle, name.encode(), which, arr.ctypes.data, arr.size))
            step = max(step, int(np.asarray(st["step"]).reshape(-1)[0]))
        cs = C.c_int32(step)
        native.check(L.cm_train_opt_step(self._handle, C.byref(cs), 1))

    def sync_trained(self):
        """Pull the trained master weights back into state_dict()."""
        L = native.lib()
        native.check(L.cm_train_sync(self._handle))
        for name, arr in self._params.items():
            out = np.empty(arr.shape, dtype=np.float32)
            native.check(L.cm_model_get_param(self._handle, name.encode(), out.ctypes.data, out.size))
            self._params[name] = out

    def mse_loss(self, pred, target) -> float:
        """F.mse_loss(pred, target) on the device (ddpm.py:120)."""
        a = native.DeviceBuffer.from_array(np.ascontiguousarray(pred, dtype=np.float32), self.device)
        b = native.DeviceBuffer.from_array(np.ascontiguousarray(target, dtype=np.float32), self.device)
        out = C.c_float()
        native.check(native.lib().cm_mse_loss(self._handle, a.ptr, b.ptr, int(np.asarray(pred).size), C.byref(out), None))
        return float(out.value)

    def debug_activation(self, name: str) -> np.ndarray:
        """Activation of the last forward by reference module name, layout [B,C,H,W,L]
        (rows beyond the last batch are stale).  Test hook."""
        L = native.lib()
        if self._handle is None:
            raise RuntimeError("no forward has run yet")
        cap = 1 << 24
        buf = np.empty(cap, dtype=np.float32)
        shape = (C.c_int64 * 5)()
        native.check(L.cm_debug_activation(self._handle, name.encode(), buf.ctypes.data, cap, shape))
        shp = tuple(int(v) for v in shape)
        return buf[: int(np.prod(shp))].reshape(shp).copy()

    def cost(self, B: int):
        f, b = C.c_double(), C.c_double()
        native.check(native.lib().cm_model_cost(self._handle, B, C.byref(f), C.byref(b)))
        return f.value, b.value

    def conv3_exec_flops(self, B: int) -> float:
        """Matrix-core FLOPs the 3x3x3 convolutions actually execute (Winograd / parity forms run fewer)."""
        fl = (C.c_double * 8)()
        native.check(native.lib().cm_model_exec_flops(self._handle, B, fl))
        return float(fl[0])

    def exec_flops(self, B: int) -> float:
        """Matrix-core FLOPs one forward executes over all kernel classes (convolutions in their reduced forms + attention)."""
        fl = (C.c_double * 8)()
        native.check(native.lib().cm_model_exec_flops(self._handle, B, fl))
        return float(sum(fl))

    def conv3_issue_flops(self, B: int):
        """(fp32-instruction FLOPs, 16-bit-operand-instruction FLOPs) the 3x3x3 convolutions ISSUE on the matrix cores: a
        six-term layer (fp32 products from exact three-way bf16 splits) issues six bf16 products per fp32-equivalent one."""
        f32, b16 = (C.c_double * 8)(), (C.c_double * 8)()
        native.check(native.lib().cm_model_issue_flops(self._handle, B, f32, b16))
        return float(f32[0]), float(b16[0])

    def conv3_flops(self, B: int) -> float:
        """Algorithmic FLOPs of the 3x3x3 convolutions of one forward at batch B."""
        fl = (C.c_double * 8)()
        native.check(native.lib().cm_model_class_flops(self._handle, B, fl))
        return float(fl[0])
