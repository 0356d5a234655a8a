"""Training-step driver for the distillation hot path: synthetic nuScenes-shaped batches, model
construction from a config, AdamW + grad-clip step, data-parallel over RCCL.

Replaces, for measurement purposes, the orchestration the reference delegates to mmcv/mmdet
(``tools/train.py`` -> ``mmdet.apis.train_detector`` -> ``EpochBasedRunner.run_iter`` ->
``MMDistributedDataParallel.train_step`` -> ``OptimizerHook``): one process per GPU,
samples sharded across ranks, the only collective is the bucketed gradient all-reduce that
``torch.nn.parallel.DistributedDataParallel`` overlaps with backward (backend "nccl" = RCCL over
xGMI).  The teacher is not a registered submodule, so DDP neither broadcasts nor reduces it
(bevdet_distill.py:1599-1610).
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import synthetic as syn
from .center_head import LiDARBoxes
from .config import Config
from . import detectors  # noqa: F401  (registers every model component)
from .registry import build_detector

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs",
                              "distillbev_centerpoint2bevdepth4d_r50.py")


def make_batch(B, rng, device, n_points=240000, n_boxes=30, n_cams=6, input_size=(256, 704), downsample=16):
    """One synthetic batch in the reference's input contract:
    img_inputs = (imgs f32[B, 2*N, 3, H, W] interleaved cur/adj per camera,
                  rots[B,2N,3,3], trans[B,2N,3], intrins[B,2N,3,3], post_rots[B,2N,3,3], post_trans[B,2N,3]
                  (first N = current frame, last N = adjacent frame), depth_gt f32[B, 2N, fH, fW]);
    points: list of f32[n_points, 5]; gt_bboxes_3d: list of LiDARBoxes (host); gt_labels_3d: list of int64."""
    H, W = input_size
    fH, fW = H // downsample, W // downsample
    cur = syn.camera_rig(B, rng, n_cams=n_cams, input_size=input_size)
    adj = {k: v.copy() for k, v in cur.items()}
    adj["trans"] = adj["trans"] + np.concatenate(
        [rng.uniform(0.0, 2.0, (B, 1, 1)), rng.uniform(-0.2, 0.2, (B, 1, 1)), np.zeros((B, 1, 1))], 2).astype(np.float32)
    mats = {k: torch.from_numpy(np.concatenate([cur[k], adj[k]], 1)).to(device) for k in cur}
    g = torch.Generator(device="cpu").manual_seed(int(rng.integers(0, 2 ** 31)))
    imgs = torch.randn((B, 2 * n_cams, 3, H, W), generator=g).to(device)
    dgt = torch.from_numpy(syn.depth_gt(B, 2 * n_cams, fH, fW, rng)).to(device)
    points, boxes, labels = [], [], []
    for _ in range(B):
        points.append(torch.from_numpy(syn.lidar_points(n_points, rng)).to(device))
        b, l = syn.gt_boxes(n_boxes, rng)
        boxes.append(LiDARBoxes(b))
        labels.append(torch.from_numpy(l))
    img_inputs = (imgs, mats["rots"], mats["trans"], mats["intrins"], mats["post_rots"], mats["post_trans"], dgt)
    return dict(points=points, img_inputs=img_inputs, gt_bboxes_3d=boxes, gt_labels_3d=labels)


def synthetic_teacher_checkpoint(model_cfg, seed=0, directory=None):
    """No trained CenterPoint weights are reachable from the build or GPU boxes (no network).  The distillation recipe
    nevertheless REQUIRES a teacher checkpoint (`inherit_head=True` asserts one, bevdet_distill.py:175-176), so the
    benchmark / tests write a seeded random-init teacher to disk in the mmdet3d checkpoint format
    ({'meta': ..., 'state_dict': {mmdet3d key names}}) and hand the PATH to the detector -- the teacher then goes through
    the same `teacher_ckpt` loading code a real epoch_20.pth would.  Returns the path."""
    import hashlib
    import tempfile
    tc = model_cfg["teacher_config"]
    if isinstance(tc, str):
        root = model_cfg.get("config_root")
        path = tc if os.path.isabs(tc) or os.path.exists(tc) or not root else os.path.join(root, tc)
        tc = Config.fromfile(path)
    tmodel = tc["model"] if "model" in tc else tc
    tag = hashlib.sha1((repr(tmodel) + str(seed)).encode()).hexdigest()[:12]
    out = os.path.join(directory or tempfile.gettempdir(), f"dbev_synthetic_teacher_{tag}_{os.getpid()}.pth")
    if not os.path.isfile(out):
        state = torch.random.get_rng_state()
        torch.manual_seed(1000 + seed)
        teacher = build_detector(tmodel)
        if hasattr(teacher, "init_weights"):        # e.g. the level embeddings of the transformer heads are allocated uninitialised
            teacher.init_weights()
        torch.random.set_rng_state(state)
        torch.save({"meta": {"note": "seeded random init (synthetic_teacher_checkpoint)", "seed": seed},
                    "state_dict": teacher.state_dict()}, out)
        import atexit
        atexit.register(lambda p=out: os.path.isfile(p) and os.remove(p))
    return out


def build_model(config=None, cfg_options=None, seed=0, allow_synthetic_teacher=False):
    """Build the detector of `config`.  A recipe that inherits the teacher's head needs `teacher_ckpt`
    (bevdet_distill.py:175-176 asserts it); `allow_synthetic_teacher=True` -- benchmarks and tests only -- substitutes a
    seeded random-init checkpoint for a missing one and says so; without it the detector's own assert fires."""
    cfg = Config.fromfile(config or DEFAULT_CONFIG) if not isinstance(config, Config) else config
    if cfg_options:
        cfg.merge_from_args(cfg_options) if isinstance(cfg_options, (list, tuple)) else cfg.merge_from_dict(cfg_options)
    m = cfg.model
    ck = m.get("teacher_ckpt")
    if (allow_synthetic_teacher and m.get("teacher_config") is not None and m.get("inherit_head")
            and not (isinstance(ck, str) and ck.lower() != "none")):
        m["teacher_ckpt"] = synthetic_teacher_checkpoint(m, seed)
        import warnings
        warnings.warn(f"build_model: no teacher_ckpt in the recipe -- distilling from a SYNTHETIC (seeded random-init) teacher, "
                      f"{m['teacher_ckpt']}", RuntimeWarning)
    torch.manual_seed(seed)
    model = build_detector(cfg.model)
    model.init_weights()
    return model, cfg


class _TrainWrapper(nn.Module):
    """DDP needs the loss computation inside forward()."""

    def __init__(self, detector):
        super().__init__()
        self.detector = detector

    def forward(self, **batch):
        return self.detector.forward_train(**batch)


def parse_losses(losses):
    """mmdet BaseDetector._parse_losses: total = sum of every entry whose key contains 'loss'.  The ~50 scalar terms of the
    distillation step are stacked (views) and added by ONE reduction instead of a chain of 0-dim additions (one launch each);
    every term's gradient is exactly 1 either way, the total differs from the sequential sum by fp32 rounding only."""
    vals = [v for k, v in losses.items() if "loss" in k]
    if len(vals) > 2 and all(torch.is_tensor(v) and v.numel() == 1 and v.dtype == vals[0].dtype and v.device == vals[0].device
                             for v in vals):
        return torch.stack([v.reshape(()) for v in vals]).sum()
    return sum(vals)


def accelerate_modules(detector):
    """Rewire parameter-free / parameter-preserving module pairs of a built, channels-last, on-GPU detector
    (student + hidden teacher) onto the gfx950 kernels: BatchNorm2d->ReLU pairs (bn_act.fuse_bn_relu_modules),
    bilinear align_corners nn.Upsample (UpsampleBilinearAC), NHWC teacher canvas.  State-dict keys are unchanged."""
    from .bn_act import fuse_bn_relu_modules
    from .distill_loss import UpsampleBilinearAC
    from .skinny_conv import use_skinny_convs
    roots = [detector] + ([detector.teacher_model] if getattr(detector, "teacher_model", None) is not None else [])
    n_bn = sum(fuse_bn_relu_modules(r) for r in roots)
    detector.skinny_convs = sum(use_skinny_convs(r) for r in roots)    # final 64 -> 1..3 convs of the CenterHead branches
    from .wino import link_conv_norm_stacks, use_wino_convs
    detector.wino_convs = sum(use_wino_convs(r) for r in roots)        # 3x3 stride-1 convolutions: Winograd F(2x2, 3x3) on fp32 MFMA
    detector.folded_norm_pairs = sum(link_conv_norm_stacks(r) for r in roots)   # conv -> eval norm -> ReLU of frozen stacks: one launch
    from .center_head import CenterHead
    from .head_batch import plan_branches
    detector.batched_branches = 0
    if os.environ.get("DBEV_HEAD_BATCH", "1") != "0":
        detector.batched_branches = sum(plan_branches(m) for r in roots for m in r.modules() if isinstance(m, CenterHead))
    from .gemm_bf6 import use_bf6_convs
    detector.bf6_convs = sum(use_bf6_convs(r) for r in roots)          # bias-free 1x1 convolutions: fp32 GEMM on the bf16 matrix cores (bf16x6)
    from .gemm_bf6 import use_bf6_linears
    detector.bf6_linears = sum(use_bf6_linears(r) for r in roots)      # nn.Linear on [tokens, C] with enough tokens (the BEVFormer encoder)
    from .stem import use_stem_convs
    detector.stem_convs = sum(use_stem_convs(r) for r in roots)        # the image backbone's 7x7 / stride-2 stem: fp32 MFMA kernel of its own
    from .colsum import use_bias_sum_convs
    detector.bias_sum_convs = sum(use_bias_sum_convs(r) for r in roots)   # remaining nn.Conv2d(bias=True): bias gradient as one streaming pass
    n_up = 0
    for r in roots:
        for mod in r.modules():
            for name, child in list(mod._modules.items()):
                if type(child) is nn.Upsample and child.mode == "bilinear" and child.align_corners \
                        and child.size is None and isinstance(child.scale_factor, (int, float)):
                    mod._modules[name] = UpsampleBilinearAC(child.scale_factor)
                    n_up += 1
        me = getattr(r, "pts_middle_encoder", None)
        if me is not None:
            me.channels_last = True            # canvas written NHWC: no NCHW->NHWC copy in front of SECOND
    return n_bn, n_up


def to_channels_last(detector):
    """Put a built, on-device detector (student + hidden teacher) into the layout bench.py times: channels-last weights of the dense
    2-D convolution modules, then `accelerate_modules`.  -> (fused norm+ReLU pairs, swapped upsample modules)"""
    # OIHW weights of the dense 2-D convolution modules only: Module.to(memory_format=) would also re-stride the sparse
    # layers' [ky, kx, Cin, Cout] / 5-D weights, whose kernels view them in their own layout
    from .dcn import ModulatedDeformConv2dPack
    for root in (detector, getattr(detector, "teacher_model", None)):
        for m in (root.modules() if root is not None else ()):
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, ModulatedDeformConv2dPack)) and m.weight.dim() == 4:
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    detector.channels_last = True
    return accelerate_modules(detector)


def _flat_view(g):
    """1-D view of a dense gradient in MEMORY order; falls back to a logical-order copy for exotic strides (the copy is
    then written back through the same mapping)."""
    if g.is_contiguous():
        return g.view(-1)
    if g.dim() == 4 and g.is_contiguous(memory_format=torch.channels_last):
        return g.permute(0, 2, 3, 1).reshape(-1)        # a view: NHWC is the physical order
    return g.reshape(-1)


class GradReducer:
    """Data-parallel gradient averaging for the one collective of the step: the student's 54 M fp32
    gradients (217 MB) in ~64 MB buckets.  Each bucket is packed with ONE concat kernel, pre-divided by the world
    size, all-reduced asynchronously (RCCL ring over xGMI on the GPU, gloo in the CPU tests) and unpacked with one
    multi-tensor copy; the buckets' collectives are in flight together.

    Why not torch's DistributedDataParallel (kept behind DBEV_TORCH_DDP=1): on MI355X its reducer costs 10 ms per
    step at this model size (per-parameter autograd hooks, 330 grad->bucket copies and divisions: measured 188 vs
    178 ms at one rank) to hide a collective that takes ~1.5-3 ms on 8 GPUs of one node -- the exposed all-reduce
    is cheaper than the machinery that overlaps it.  Semantics are DDP's: parameters (and buffers, once) are
    broadcast from rank 0 at construction, every rank ends the step with identical averaged gradients and BatchNorm
    statistics stay local.

    Parameters without a gradient (mmcv's `find_unused_parameters`; e.g. the 'backbone*' adaptation layers while
    `_epoch < multi_scale_epoch`, bevdet_distill.py:1452-1455): a rank that has no gradient for a parameter
    contributes zeros, one flag per parameter rides in the bucket, and a parameter that NO rank produced a gradient
    for keeps `grad = None` (the optimizer skips it, as under DDP) -- ranks never block each other.

    Layout: a gradient is packed in the PARAMETER's memory layout (a gradient that arrives with other strides, e.g. a
    permuted view out of a custom backward, is first copied into that layout), so every rank packs every element at
    the same bucket offset whatever kernels produced its gradients.
    """

    def __init__(self, params, buffers=(), bucket_mb=64, overlap=None):
        self.params = list(params)
        self.world = dist.get_world_size()
        self.host_staged = dist.get_backend() == "gloo"      # gloo: device gradients travel through a host buffer
        with torch.no_grad():
            for t in list(self.params) + list(buffers):
                if self.host_staged and t.is_cuda:
                    h = t.detach().cpu()
                    dist.broadcast(h, 0)
                    t.copy_(h)                     # in place on the tensor itself (not .data): the version counter moves
                else:
                    dist.broadcast(t.detach(), 0)  # the collective writes through a raw pointer ...
                    torch.autograd.graph.increment_version(t)   # ... so everything keyed on _version (bn_act's kept eval coefficients) is told
        cap = int(bucket_mb) * (1 << 20)
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):            # gradients become ready roughly in reverse parameter order
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= cap:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        # Overlap with backward (mmcv's MMDistributedDataParallel does it per 25 MB bucket, tools/distributed.py:29-79): a
        # post-accumulate-grad hook per parameter marks it ready; a bucket whose parameters are all ready is packed and its
        # all-reduce started AT ONCE, from inside backward -- but strictly in bucket order, so every rank issues the same sequence
        # of collectives whatever order (or subset) of gradients its own batch produced.  Buckets holding a parameter without a
        # gradient on this rank simply wait for all_reduce_grads(), which flushes the rest in the same order.
        # Default OFF (round 5): the ordering / bit-identity of the overlapped path is covered by the 2-rank tests, but whether launching
        # the collectives from the autograd thread helps on xGMI has never been measured on N > 1 GPUs (the 8-GPU node has not been
        # available to this project) -- the exposed all-reduce of 217 MB is ~1-2 % of the step, the simple path is the default until a
        # measurement says otherwise.  DBEV_DDP_OVERLAP=1 (or overlap=True) turns it on.
        self.overlap = (os.environ.get("DBEV_DDP_OVERLAP", "0") == "1") if overlap is None else bool(overlap)
        self._bucket_of = {id(p): i for i, bucket in enumerate(self.buckets) for p in bucket}
        self._seen = [set() for _ in self.buckets]
        self._fired, self._pending, self.fired_in_backward = 0, [], 0
        self._flags = {}                  # (bucket index, have pattern) -> device flag row, built once (no per-step upload)
        self._accumulating = False        # inside no_sync(): hooks do nothing
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params] if self.overlap else []

    def close(self):
        """detach from the parameters: remove the autograd hooks and drop anything in flight (a discarded reducer must not keep
        launching collectives from another reducer's / trainer's backward)"""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.reset()

    def dirty(self):
        """state left behind by a step that did not reach all_reduce_grads() (an exception between backward and the reduction, a
        loop that skips a step on a non-finite loss)"""
        return bool(self._pending) or self._fired > 0 or any(self._seen)

    def reset(self):
        """forget an aborted step: wait for the collectives it started (every rank started the same ones -- they were issued in
        bucket order), drop their results, clear the per-bucket bookkeeping.  Trainer.step() calls it before every forward, so a
        step that raised (or was skipped) after its backward cannot poison the next one; every rank must skip the same steps."""
        for work, *_ in self._pending:
            work.wait()
        self._pending = []
        self._fired = 0
        for s_ in self._seen:
            s_.clear()

    class _NoSync:
        def __init__(self, r):
            self.r = r

        def __enter__(self):
            self.prev, self.r._accumulating = self.r._accumulating, True

        def __exit__(self, *exc):
            self.r._accumulating = self.prev

    def no_sync(self):
        """gradient accumulation (DDP's ``no_sync``): backward passes inside the context only accumulate into ``p.grad``; the ONE
        backward outside it -- followed by all_reduce_grads() -- reduces the accumulated sum.  Without it a second backward before
        all_reduce_grads() raises (its gradients would arrive after their bucket has been packed)."""
        return GradReducer._NoSync(self)

    @staticmethod
    def _in_param_layout(p, g):
        if g.shape == p.shape and g.stride() == p.stride():
            return g
        out = torch.empty_like(p)                  # preserve_format: p's strides
        out.copy_(g)
        return out

    def _on_grad(self, p):
        if self._accumulating:
            return
        i = self._bucket_of[id(p)]
        if i < self._fired or id(p) in self._seen[i]:
            # the bucket was packed (or the parameter counted) by an earlier backward of this step: the state machine is one
            # backward per all_reduce_grads() -- anything else would silently average a stale bucket
            raise RuntimeError("GradReducer: a second gradient arrived for a parameter before all_reduce_grads(); wrap the extra "
                               "backward passes of a gradient-accumulation step in reducer.no_sync(), or call reducer.reset() after "
                               "abandoning a step between its backward and all_reduce_grads()")
        self._seen[i].add(id(p))
        if i == self._fired:
            n0 = self._fired
            self._fire(force=False)
            self.fired_in_backward += self._fired - n0

    @torch.no_grad()
    def _fire(self, force):
        while self._fired < len(self.buckets) and (force or len(self._seen[self._fired]) == len(self.buckets[self._fired])):
            self._launch(self.buckets[self._fired])
            self._fired += 1

    def _flag_row(self, bucket, have, like):
        """per-parameter "this rank has a gradient" flags (x world, the pre-division brings them back to 0 / 1) as a device row; kept
        per (bucket, pattern): a training run sees one or two patterns per bucket, so after the first step there is no upload at all
        -- the first one goes through pinned staging (`_lib.h2d`): a pageable copy issued from the autograd thread would block the host
        until the GPU queue drains, once per bucket, inside backward"""
        key = (id(bucket), tuple(have))
        row = self._flags.get(key)
        if row is None:
            host = torch.tensor([float(h) * self.world for h in have], dtype=like.dtype)
            if like.is_cuda:
                from ._lib import h2d
                row = h2d(host, like.device)
            else:
                row = host
            self._flags[key] = row
        return row

    def _launch(self, bucket):
        """pack one bucket (ONE concat kernel over memory-order views + the per-parameter flags), pre-divide, start its all-reduce"""
        grads, have = [], []
        for p in bucket:
            have.append(p.grad is not None)
            g = self._in_param_layout(p, p.grad) if p.grad is not None else torch.zeros_like(p)
            if p.grad is not None and g is not p.grad:
                p.grad = g
            grads.append(g)
        flats = [_flat_view(g) for g in grads]          # memory-order 1-D views: no per-tensor copy kernels
        flat = torch.cat(flats + [self._flag_row(bucket, have, flats[0])])
        flat.div_(self.world)
        host = flat.cpu() if self.host_staged and flat.is_cuda else None
        work = dist.all_reduce(host if host is not None else flat, op=dist.ReduceOp.SUM, async_op=True)
        self._pending.append((work, flat, flats, grads, bucket, have, host))

    @torch.no_grad()
    def all_reduce_grads(self):
        """after backward: start whatever has not been started, wait, hand the averaged gradients back"""
        self._fire(force=True)
        for work, flat, flats, grads, bucket, have, host in self._pending:
            work.wait()
            if host is not None:
                flat.copy_(host)
            n = len(flats)
            parts = list(flat.split([f.numel() for f in flats] + [n]))
            any_rank = None if all(have) else (parts[n].cpu() > 0).tolist()     # rare path: one host read-back
            for j, (p, f, g, part) in enumerate(zip(bucket, flats, grads, parts[:n])):
                if not have[j] and not any_rank[j]:
                    continue                                # NO rank produced a gradient: stays None (the optimizer skips it)
                if f.data_ptr() == g.data_ptr():
                    # the averaged gradient stays where the collective left it: a view of the bucket with the parameter's
                    # strides (no 217 MB copy-back); the inverse of _flat_view's two memory-order layouts
                    if g.is_contiguous():
                        p.grad = part.view(g.shape)
                    else:
                        N, C, H, W = g.shape
                        p.grad = part.view(N, H, W, C).permute(0, 3, 1, 2)
                else:                                       # exotic parameter strides: the flat piece was a logical-order copy
                    g.copy_(part.view(g.shape))
                    p.grad = g
        self._pending = []
        self._fired = 0
        for s_ in self._seen:
            s_.clear()


def param_groups(detector, opt):
    """mmcv DefaultOptimizerConstructor for ``paramwise_cfg=dict(custom_keys={substring: dict(lr_mult=, decay_mult=)})`` (the
    BEVFormer recipes train the image backbone at lr x 0.1): pops ``paramwise_cfg`` from ``opt`` and returns the optimizer's
    parameter groups -- the longest matching key wins, as in mmcv."""
    pw = opt.pop("paramwise_cfg", None) or {}
    keys = sorted(pw.get("custom_keys", {}), key=lambda k: (-len(k), k))
    named = [(n, p) for n, p in detector.named_parameters() if p.requires_grad]
    if not keys:
        return [p for _, p in named]
    groups = {}
    for n, p in named:
        hit = next((k for k in keys if k in n), None)
        groups.setdefault(hit, []).append(p)
    out = []
    for k, ps in groups.items():
        g = dict(params=ps)
        if k is not None:
            c = pw["custom_keys"][k]
            g["lr"] = opt["lr"] * c.get("lr_mult", 1.0)
            if opt.get("weight_decay") is not None:
                g["weight_decay"] = opt["weight_decay"] * c.get("decay_mult", 1.0)
        out.append(g)
    return out


class Trainer:
    def __init__(self, model, cfg, device, world_size=1, channels_last=False):
        from .miopen_tuning import use_shipped_db
        use_shipped_db()                     # MIOpen solver tables tuned on MI355X; before the process's first convolution
        self.device = device
        self.detector = model.to(device)
        self.detector.train()
        self.packer = None
        if channels_last:
            self.fused_bn_relu, self.swapped_upsample = to_channels_last(self.detector)
            if device.type == "cuda":
                from .packer import WeightPacker
                self.packer = WeightPacker([self.detector])      # the trainable layers' packed weights: one launch per family and step
                from . import graphed
                if graphed.enabled(world_size) and hasattr(self.detector, "image_encoder"):
                    # the gradient-free adjacent frame's backbone + neck as one hipGraph.  It re-validates every derived weight buffer
                    # it baked in before each replay (graphed.py); with the packs above refreshed in place after every optimizer
                    # step that check finds everything fresh
                    det = self.detector
                    mods = [getattr(det, "img_backbone", None), getattr(det, "img_neck", None)]
                    norms = lambda: [mod for m in mods if m is not None for mod in m.modules()
                                     if isinstance(mod, nn.modules.batchnorm._BatchNorm)]
                    det.adjacent_graph = graphed.GraphedNoGrad(det.image_encoder, token=lambda: graphed.state_token(*mods), norms=norms)
        self.wrapper = _TrainWrapper(self.detector)
        self.world_size = world_size
        self.reducer = None
        distributed = world_size > 1 or os.environ.get("DBEV_FORCE_DDP") == "1"
        if distributed and os.environ.get("DBEV_TORCH_DDP") != "1":
            # default data-parallel path: bucketed flat gradient all-reduce right after backward (GradReducer)
            self.module = self.wrapper
            self.reducer = GradReducer([p for p in self.detector.parameters() if p.requires_grad],
                                       list(self.detector.buffers()), bucket_mb=32)
        elif distributed:
            self.module = nn.parallel.DistributedDataParallel(
                self.wrapper, device_ids=[device.index] if device.type == "cuda" else None, broadcast_buffers=False,
                find_unused_parameters=False, gradient_as_bucket_view=True, bucket_cap_mb=64)
            # one pre-division per 64 MB bucket instead of DDP's built-in per-parameter div_ (270 tiny launches
            # per step, 2.3 ms on MI355X); same arithmetic: grad / world, then the RCCL sum all-reduce
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
            self.module.register_comm_hook(None, default_hooks.allreduce_hook)
        else:
            self.module = self.wrapper
        opt = dict(cfg.get("optimizer", dict(type="AdamW", lr=2e-4, weight_decay=0.01)))
        assert opt.pop("type") == "AdamW"
        params = [p for p in self.detector.parameters() if p.requires_grad]
        # (round 5's one-launch AdamW with the clip factor inside -- bit-identical, 0 ms gained in alternating A/B runs -- was retired in
        # round 6: docs/design/09_round6.md)
        self.optimizer = torch.optim.AdamW(param_groups(self.detector, opt), **opt, fused=(device.type == "cuda"))
        oc = cfg.get("optimizer_config", {}) or {}
        gc = oc.get("grad_clip", None)
        self.grad_clip = dict(gc) if gc else None
        self.params = params

    def close(self):
        """drop the data-parallel reducer's autograd hooks and the captured graph (a Trainer that is discarded while its detector lives on)"""
        if self.reducer is not None:
            self.reducer.close()
            self.reducer = None
        if getattr(self.detector, "adjacent_graph", None) is not None:
            self.detector.adjacent_graph = None      # (installed by this trainer: goes with it)

    def step(self, batch):
        if self.reducer is not None and self.reducer.dirty():
            self.reducer.reset()                 # the previous step was abandoned after (part of) its backward
        losses = self.module(**batch)
        loss = parse_losses(losses)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if self.reducer is not None:
            self.reducer.all_reduce_grads()
        if self.grad_clip:
            nn.utils.clip_grad_norm_(self.params, **self.grad_clip)
        self.optimizer.step()
        if self.packer is not None:
            self.packer.repack()                 # (after the step's version bump: the layers find their packs fresh in the next forward)
        return loss.detach(), losses
