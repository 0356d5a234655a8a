"""AdamW whose step is ONE launch over all parameter tensors (csrc/adamw.hip), with the gradient-clipping factor applied as the gradient
is read.  Replaces the pair `torch.nn.utils.clip_grad_norm_` (its scaling pass) + `torch.optim.AdamW.step` that mmcv's OptimizerHook
runs for the reference's recipes (optimizer = AdamW(lr=2e-4, weight_decay=0.01), grad_clip = dict(max_norm=35)).

`MultiTensorAdamW` IS a torch.optim.AdamW (fused=True): same constructor, same state (`step` on the device, `exp_avg`, `exp_avg_sq`),
same state_dict -- a checkpoint of either loads into the other.  Only `step()` differs: when every parameter group's tensors are fp32,
on one GPU, dense and laid out like their gradients and moments, each group is one dbev_adamw_multi launch; anything else (amsgrad,
maximize, capturable, a sparse / differently-strided gradient, tensors of one group at different step counts) makes the whole step
torch's own fused one.  `step(grad_scale=c)`: c is a 0-dim device
tensor every gradient value is multiplied by (the clip factor); the gradients themselves are NOT rewritten -- `p.grad` still holds
the unclipped values after the step (DBEV_FUSED_CLIP=0 in train_step.Trainer: clip in place as before)."""
import math
import os

import numpy as np
import torch

from . import _lib as L

_ON = os.environ.get("DBEV_ADAMW_KERNEL", "1") != "0"       # (train_step.Trainer installs this optimizer only with DBEV_ADAMW=1)

_TENSOR = np.dtype([("p", np.uint64), ("g", np.uint64), ("m", np.uint64), ("v", np.uint64), ("n", np.int64)])
assert _TENSOR.itemsize == 40


def _dense_like(p, *others):
    if not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
        return False
    return all(o.dtype == torch.float32 and o.device == p.device and o.shape == p.shape and o.stride() == p.stride() for o in others)


class MultiTensorAdamW(torch.optim.AdamW):
    def __init__(self, params, **kw):
        kw.setdefault("fused", True)
        super().__init__(params, **kw)
        self._plans = {}               # group index -> (signature, device chunk map, n chunks, host pointer tables (2), flip)
        self.multi_launches = 0
        self.torch_steps = 0

    def _group_ok(self, group, params):
        if not (_ON and not group["amsgrad"] and not group["maximize"] and not group.get("capturable", False)
                and not group.get("differentiable", False) and not isinstance(group["lr"], torch.Tensor)
                and all(p.is_cuda and p.dtype == torch.float32 and not p.grad.is_sparse for p in params)
                and len({p.device for p in params}) == 1):
            return False
        for p in params:                                          # state as torch's _init_group makes it for fused=True
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return all(_dense_like(p, p.grad, self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for p in params)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        work = [(gi, g, [p for p in g["params"] if p.grad is not None]) for gi, g in enumerate(self.param_groups)]
        work = [w for w in work if w[2]]
        if not all(self._ready(gi, g, ps) for gi, g, ps in work):
            # torch's own fused step for everything (the factor applied in place first, as clip_grad_norm_ does)
            if grad_scale is not None:
                torch._foreach_mul_([p.grad for _, _, ps in work for p in ps], grad_scale)
            self._plans.clear()
            self.torch_steps += 1
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group, params in work:
            self._multi_step(gi, group, params, grad_scale)
        return loss

    def _ready(self, gi, group, params):
        """can this group take the one-launch step now?  The full checks (dtype, device, dense layout shared by parameter, gradient and
        moments, one step count) run when the group's plan is made; a step with the same parameter storage only re-checks what is new
        every step -- the gradient tensors' layout (the end of a step is host-latency-bound: ~0.4 us per tensor, not 2)"""
        plan = self._plans.get(gi)
        if plan is not None and len(plan["strides"]) == len(params):
            ok = True
            for p, st, ptr in zip(params, plan["strides"], plan["key"]):
                g = p.grad
                if p.data_ptr() != ptr or g.stride() != st or g.dtype != torch.float32 or g.is_sparse:
                    ok = False
                    break
            if ok:
                return plan["step"] is not None
        return self._group_ok(group, params) and self._plan(gi, group, params)

    def _plan(self, gi, group, params):
        """chunk map + pointer tables of one group (rebuilt when a tensor changes); False: the tensors of the group stand at different
        step counts (torch's per-tensor bias corrections are needed)"""
        dev = params[0].device
        key = tuple(p.data_ptr() for p in params)
        plan = None
        if True:
            sig = tuple((p.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(), p.numel()) for p in params)
            ce = int(L.lib().dbev_adamw_chunk_elems())
            counts = np.array([(p.numel() + ce - 1) // ce for p in params], dtype=np.int64)
            t_idx = np.repeat(np.arange(len(params), dtype=np.int32), counts)
            c_idx = np.concatenate([np.arange(c, dtype=np.int32) for c in counts])
            chunks = np.stack([t_idx, c_idx], 1).astype(np.int32)
            host = []
            for _ in range(2):                                    # two pinned tables: the previous step's copy may still be in flight
                h = torch.empty((len(params) * _TENSOR.itemsize,), dtype=torch.uint8).pin_memory()
                a = h.numpy().view(_TENSOR)
                a["p"] = [s[0] for s in sig]; a["m"] = [s[1] for s in sig]; a["v"] = [s[2] for s in sig]; a["n"] = [s[3] for s in sig]
                host.append((h, a))
            # the step counter the bias corrections need on the HOST: read once (a checkpoint may have put any value there)
            steps = {float(self.state[p]["step"].item()) for p in params}
            plan = {"key": key, "strides": [p.stride() for p in params], "chunks": L.h2d(torch.from_numpy(chunks.reshape(-1).view(np.uint8).copy()), dev), "n": int(chunks.shape[0]),
                    "host": host, "flip": 0, "dev_table": torch.empty((len(params) * _TENSOR.itemsize,), dtype=torch.uint8, device=dev),
                    "step": steps.pop() if len(steps) == 1 else None, "steps": [self.state[p]["step"] for p in params]}
            self._plans[gi] = plan
        return plan["step"] is not None

    def _multi_step(self, gi, group, params, grad_scale):
        dev = params[0].device
        plan = self._plans[gi]
        plan["step"] += 1.0
        torch._foreach_add_(plan["steps"], 1.0)                   # the state's own counters (checkpoints, torch's step after a fall-back)
        beta1, beta2 = group["betas"]
        bc1 = float(np.float32(1.0 - beta1 ** plan["step"]))
        bc2s = float(np.float32(math.sqrt(1.0 - beta2 ** plan["step"])))
        h, a = plan["host"][plan["flip"]]
        plan["flip"] ^= 1
        a["g"] = [p.grad.data_ptr() for p in params]
        plan["dev_table"].copy_(h, non_blocking=True)
        with torch.cuda.device(dev):
            L.call("dbev_adamw_multi", L.ptr(plan["dev_table"]), L.ptr(plan["chunks"]), plan["n"], L.ptr(grad_scale), float(group["lr"]),
                   float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]), bc1, bc2s, L.stream_ptr(dev))
        self.multi_launches += 1

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans.clear()                                       # new moment tensors, new step counts


def clip_factor(params, max_norm, norm_type=2.0, error_if_nonfinite=False, foreach=None):
    """(total_norm, c): the factor torch.nn.utils.clip_grad_norm_ would multiply every gradient by -- min(1, max_norm / (norm + 1e-6)) --
    as a 0-dim device tensor, without touching the gradients"""
    grads = [p.grad for p in params if p.grad is not None]
    total = torch.nn.utils.get_total_norm(grads, norm_type, error_if_nonfinite, foreach)
    c = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, c
