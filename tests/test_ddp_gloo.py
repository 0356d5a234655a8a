"""CPU, world_size 2, gloo: the data-parallel path of the training step (Trainer + GradReducer, and the
torch DistributedDataParallel alternative behind DBEV_TORCH_DDP=1):
samples are sharded across ranks, the only collective is the gradient all-reduce, the hidden
teacher is neither broadcast nor reduced, and N ranks x B samples == one process x N*B samples."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from distill_bev_amd.config import Config
from distill_bev_amd.train_step import Trainer, parse_losses


class ToyDetector(nn.Module):
    """Pure-torch stand-in with the detector contract: forward_train(**batch) -> dict of losses,
    teacher hidden from parameters() exactly like BEVDepth4DDistill does it."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.net = nn.Sequential(nn.Linear(6, 16), nn.ReLU(), nn.Linear(16, 3))
        object.__setattr__(self, "teacher_model", nn.Linear(6, 3))
        for p in self.teacher_model.parameters():
            p.requires_grad_(False)

    def forward_train(self, x=None, y=None):
        out = self.net(x)
        with torch.no_grad():
            t = self.teacher_model(x)
        return {"loss_task": ((out - y) ** 2).mean(), "kd_loss": 0.1 * ((out - t) ** 2).mean(), "aux_metric": out.mean()}


CFG = Config(dict(optimizer=dict(type="AdamW", lr=1e-2, weight_decay=0.01),
                  optimizer_config=dict(grad_clip=dict(max_norm=5, norm_type=2))))


def _data(n):
    g = torch.Generator().manual_seed(5)
    return torch.randn((n, 6), generator=g), torch.randn((n, 3), generator=g)


def _worker(rank, world, port, out, torch_ddp=False, desync=False):
    os.environ["DBEV_TORCH_DDP"] = "1" if torch_ddp else "0"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y = _data(8 * world)
    det = ToyDetector()
    if desync and rank == 1:                           # ranks start from different weights: rank 0's must win
        with torch.no_grad():
            for p in det.net.parameters():
                p.add_(1.0)
    tr = Trainer(det, CFG, torch.device("cpu"), world_size=world)
    assert (tr.reducer is None) == torch_ddp
    shard = slice(rank * 8, (rank + 1) * 8)            # contiguous per-rank slices (samplers/distributed_sampler.py:35-39)
    for _ in range(3):
        loss, losses = tr.step(dict(x=x[shard], y=y[shard]))
    flat = torch.cat([p.detach().reshape(-1) for p in tr.params])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save({"params": gathered, "nparams": len(tr.params)}, out)
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("torch_ddp,desync", [(False, False), (False, True), (True, False)])
def test_two_rank_gloo_step_equals_single_process_big_batch(tmp_path, torch_ddp, desync):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_worker, args=(2, port, out, torch_ddp, desync), nprocs=2, join=True)
    res = torch.load(out)
    assert torch.equal(res["params"][0], res["params"][1])          # ranks stay in lock step
    assert res["nparams"] == 4                                       # teacher parameters are not trained / reduced
    x, y = _data(16)
    tr = Trainer(ToyDetector(), CFG, torch.device("cpu"), world_size=1)
    for _ in range(3):
        tr.step(dict(x=x, y=y))
    flat = torch.cat([p.detach().reshape(-1) for p in tr.params])
    assert torch.allclose(flat, res["params"][0], atol=1e-6)


def test_parse_losses_sums_only_loss_keys():
    d = {"loss_a": torch.tensor(1.0), "kd_fg_feat_loss_head_head": torch.tensor(2.0), "acc": torch.tensor(100.0)}
    assert float(parse_losses(d)) == 3.0


def _reducer_buckets(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distill_bev_amd.train_step import GradReducer
    torch.manual_seed(0)
    params = [nn.Parameter(torch.randn(n)) for n in (300_000, 10, 200_000, 70_000, 5, 600_000)]
    red = GradReducer(params, bucket_mb=1)                       # 1 MiB = 262 144 floats
    ids = [id(p) for b in red.buckets for p in b]
    sizes = [sum(p.numel() for p in b) for b in red.buckets]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(i + 1))
    red.all_reduce_grads()
    ok_vals = all(torch.equal(p.grad, torch.full_like(p, float(i + 1))) for i, p in enumerate(params))
    params[1].grad = None                                        # no gradient anywhere: stays None, nothing raised
    red.all_reduce_grads()
    raised = params[1].grad is not None
    torch.save(dict(ids=ids, expect=[id(p) for p in reversed(params)], sizes=sizes, ok_vals=ok_vals, raised=raised), out)
    dist.destroy_process_group()


def test_grad_reducer_buckets_cover_every_parameter_once_in_reverse_order(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "red.pt")
    mp.spawn(_reducer_buckets, args=(1, port, out), nprocs=1, join=True)
    r = torch.load(out)
    assert r["ids"] == r["expect"]                               # every parameter once, last parameter first
    assert all(sz * 4 >= (1 << 20) for sz in r["sizes"][:-1])    # all buckets but the last reach the cap
    assert r["ok_vals"]                                          # world 1: average == own gradient, packed/unpacked exactly
    assert not r["raised"]                                       # a parameter nobody has a gradient for keeps grad=None


def _reducer_unused_and_layout(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distill_bev_amd.train_step import GradReducer
    torch.manual_seed(0)
    params = [nn.Parameter(torch.randn(4, 3, 2, 2)), nn.Parameter(torch.randn(7)), nn.Parameter(torch.randn(5)),
              nn.Parameter(torch.randn(6, 4, 3, 3).contiguous(memory_format=torch.channels_last))]
    red = GradReducer(params, bucket_mb=1)
    g0 = torch.arange(48.0).view(4, 3, 2, 2) * (rank + 1)
    # the same logical gradient arrives with different strides on the two ranks (a permuted view on rank 1)
    params[0].grad = g0 if rank == 0 else g0.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    params[1].grad = torch.full((7,), 2.0) if rank == 0 else None     # used on rank 0 only
    params[2].grad = None                                            # used nowhere
    g3 = torch.arange(216.0).view(6, 4, 3, 3) * (rank + 1)
    params[3].grad = g3.contiguous(memory_format=torch.channels_last) if rank == 0 else g3.contiguous()
    red.all_reduce_grads()
    res = dict(p0=params[0].grad.clone(), p1=params[1].grad, p2=params[2].grad, p3=params[3].grad.clone(),
               p3_stride=params[3].grad.stride() == params[3].stride())
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        torch.save(gathered, out)
    dist.destroy_process_group()


def test_grad_reducer_unused_parameters_and_mixed_gradient_layouts(tmp_path):
    """world 2: a parameter unused on ONE rank receives the average of (grad, 0) on both ranks; a parameter unused on
    ALL ranks keeps grad=None; gradients that arrive with rank-dependent strides are averaged element by element."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "red2.pt")
    mp.spawn(_reducer_unused_and_layout, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out, weights_only=False)
    g0 = torch.arange(48.0).view(4, 3, 2, 2)
    g3 = torch.arange(216.0).view(6, 4, 3, 3)
    for r in (r0, r1):
        assert torch.equal(r["p0"], g0 * 1.5) and torch.equal(r["p3"], g3 * 1.5) and r["p3_stride"]
        assert torch.equal(r["p1"], torch.full((7,), 1.0))
        assert r["p2"] is None


# ---- the set-prediction heads' one collective: the averaging factors of the losses (configs[4]) ---------------------------
def _head_loss_worker(rank, world, port, out):
    """BEVFormerHead.loss on two ranks with different ground truth: `sync_cls_avg_factor` / `num_total_pos` go through
    reduce_mean (bevformer_head.py:353-364), so each rank's losses use the MEAN number of matched queries over the ranks."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import bevformer_cfgs as C
    from distill_bev_amd import bevformer  # noqa: F401
    from distill_bev_amd.center_head import LiDARBoxes
    from distill_bev_amd.registry import build_head
    from distill_bev_amd import synthetic as syn
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    head = build_head(C.small_bevformer_head_cfg())
    head.init_weights()
    g = torch.Generator().manual_seed(100 + rank)
    outs = dict(all_cls_scores=torch.randn((2, 1, 12, 10), generator=g), all_bbox_preds=torch.randn((2, 1, 12, 10), generator=g),
                enc_cls_scores=None, enc_bbox_preds=None)
    n_gt = (2, 7)[rank]
    bx, lb = syn.gt_boxes(n_gt, np.random.default_rng(rank))
    losses = head.loss([LiDARBoxes(bx)], [torch.from_numpy(lb)], outs)
    out.put((rank, {k: float(v) for k, v in losses.items()}, n_gt))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_set_prediction_loss_averages_its_factors_over_the_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = [ctx.Process(target=_head_loss_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, l, n = q.get(timeout=300)
        got[r] = (l, n)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process values of the same two shards (factor = own matched count) rescale to the two-rank ones by n_r / mean(n)
    solo = dict()
    for r in range(2):
        qq = ctx.Queue()
        _head_loss_worker(r, 1, 0, qq)
        solo[r] = qq.get()[1]
    mean_n = (got[0][1] + got[1][1]) / 2
    for r in range(2):
        for k, v in got[r][0].items():
            want = solo[r][k] * got[r][1] / mean_n
            assert abs(v - want) <= 1e-5 * abs(want), (r, k, v, want)


# ---- overlap of the bucket all-reduces with backward --------------------------------------------------------------------------
def _overlap_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distill_bev_amd.train_step import GradReducer
    res = {}
    for overlap in (False, True):
        torch.manual_seed(0)
        net = nn.Sequential(*[nn.Linear(64, 64) for _ in range(6)], nn.Linear(64, 1))
        side = nn.Linear(64, 1)                        # used on rank 1 only: its bucket cannot start inside rank 0's backward
        params = list(net.parameters()) + list(side.parameters())
        red = GradReducer(params, bucket_mb=0.02, overlap=overlap)          # ~5 buckets of one or two layers
        x = torch.randn(16, 64, generator=torch.Generator().manual_seed(10 + rank))
        fired = []
        for step in range(2):
            for p in params:
                p.grad = None
            h = net[:6](x)
            loss = net[6](h).square().mean() + (side(h).mean() if rank == 1 else 0.0)
            loss.backward()
            fired.append(red.fired_in_backward)
            red.all_reduce_grads()
        res[overlap] = dict(grads=[p.grad.clone() for p in params], fired=fired, n_buckets=len(red.buckets))
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        torch.save(gathered, out)
    dist.destroy_process_group()


def test_bucket_all_reduces_start_inside_backward_in_bucket_order(tmp_path):
    """GradReducer(overlap=True): buckets whose gradients are complete start their all-reduce from the post-accumulate-grad hooks,
    in bucket order on every rank even when the ranks' used-parameter sets differ (no deadlock, no mismatched collectives);
    the averaged gradients equal the post-backward path's bit for bit and are identical on both ranks."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "ov.pt")
    mp.spawn(_overlap_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out, weights_only=False)
    for r in (r0, r1):
        assert r[False]["fired"] == [0, 0]
        assert r[True]["n_buckets"] >= 4
        assert all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(r[False]["grads"], r[True]["grads"]))
    assert all(torch.equal(a, b) for a, b in zip(r0[True]["grads"], r1[True]["grads"]))
    # rank 1 produced every gradient: all but (at most) the last bucket started inside backward; rank 0 lacks `side`'s gradient
    # (the FIRST bucket: parameters are bucketed in reverse order), so nothing can start early there -- and nothing hangs
    assert r1[True]["fired"][1] - r1[True]["fired"][0] >= r1[True]["n_buckets"] - 1
    assert r0[True]["fired"] == [0, 0]


# ---- the hook state machine: one backward per all_reduce_grads(), no_sync() for accumulation, close() -------------------------
def _state_machine_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distill_bev_amd.train_step import GradReducer
    torch.manual_seed(0)
    net = nn.Sequential(*[nn.Linear(32, 32) for _ in range(4)], nn.Linear(32, 1))
    params = list(net.parameters())
    red = GradReducer(params, bucket_mb=0.005, overlap=True)
    xs = [torch.randn(8, 32, generator=torch.Generator().manual_seed(20 + 2 * rank + i)) for i in range(2)]
    res = {}
    # (a) a second backward before all_reduce_grads() is an error, not a silently stale bucket
    net(xs[0]).square().mean().backward()
    try:
        net(xs[1]).square().mean().backward()
        res["second_backward_raised"] = False
    except RuntimeError as e:
        res["second_backward_raised"] = "no_sync" in str(e)
    red.all_reduce_grads()
    # (b) accumulation under no_sync(): the reduced gradient is the average over ranks of the SUM of both micro-batches
    for p in params:
        p.grad = None
    with red.no_sync():
        net(xs[0]).square().mean().backward()
    assert red.fired_in_backward == res.get("_f", red.fired_in_backward)
    net(xs[1]).square().mean().backward()
    red.all_reduce_grads()
    res["accum"] = [p.grad.clone() for p in params]
    # reference: plain autograd on each rank, averaged with one all_reduce per tensor
    for p in params:
        p.grad = None
    red.close()
    res["hooks_after_close"] = len(red._hooks)
    (net(xs[0]).square().mean() + net(xs[1]).square().mean()).backward()
    net(xs[0]).square().mean().backward()              # would raise if a hook were still attached
    ref = []
    for p in params:
        p.grad = None
    (net(xs[0]).square().mean() + net(xs[1]).square().mean()).backward()
    for p in params:
        g = p.grad.clone() / world
        dist.all_reduce(g)
        ref.append(g)
    res["ref"] = ref
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        torch.save(gathered, out)
    dist.destroy_process_group()


def test_grad_reducer_rejects_a_second_backward_and_accumulates_under_no_sync(tmp_path):
    """the overlap hooks assume one backward per all_reduce_grads().  A second one now raises (instead of being averaged
    into nothing), `no_sync()` gives DDP's accumulation semantics, and `close()` detaches the hooks."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "sm.pt")
    mp.spawn(_state_machine_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out, weights_only=False)
    for r in (r0, r1):
        assert r["second_backward_raised"] is True
        assert r["hooks_after_close"] == 0
        for a, b in zip(r["accum"], r["ref"]):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    assert all(torch.equal(a, b) for a, b in zip(r0["accum"], r1["accum"]))


# ---- an abandoned step (backward ran, all_reduce_grads() did not) must not poison the next one -----------------------------------
def _abort_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from distill_bev_amd.train_step import GradReducer
    torch.manual_seed(0)
    net = nn.Sequential(*[nn.Linear(32, 32) for _ in range(4)], nn.Linear(32, 1))
    params = list(net.parameters())
    red = GradReducer(params, bucket_mb=0.005, overlap=True)
    xs = [torch.randn(8, 32, generator=torch.Generator().manual_seed(40 + 2 * rank + i)) for i in range(2)]
    res = {}
    net(xs[0]).square().mean().backward()        # step 1: backward started collectives from the hooks ... and is abandoned here
    res["dirty_after_abort"] = red.dirty()
    res["pending_after_abort"] = len(red._pending)
    red.reset()                                  # what Trainer.step() does at the start of the next step
    res["dirty_after_reset"] = red.dirty()
    for p in params:
        p.grad = None
    net(xs[1]).square().mean().backward()        # step 2 must neither raise nor average stale buckets
    red.all_reduce_grads()
    res["grads"] = [p.grad.clone() for p in params]
    red.close()
    ref = []
    for p in params:
        p.grad = None
    net(xs[1]).square().mean().backward()
    for p in params:
        g = p.grad.clone() / world
        dist.all_reduce(g)
        ref.append(g)
    res["ref"] = ref
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        torch.save(gathered, out)
    dist.destroy_process_group()


def test_grad_reducer_reset_after_an_abandoned_step(tmp_path):
    """a step that stops between backward and all_reduce_grads() (an OOM the loop catches, a skipped non-finite loss)
    leaves fired buckets and pending collectives behind; `reset()` (called by Trainer.step before every forward when `dirty()`)
    waits for and drops them, and the next step reduces exactly its own gradients."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "abort.pt")
    mp.spawn(_abort_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out, weights_only=False)
    for r in (r0, r1):
        assert r["dirty_after_abort"] is True and r["pending_after_abort"] >= 1 and r["dirty_after_reset"] is False
        for a, b in zip(r["grads"], r["ref"]):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    assert all(torch.equal(a, b) for a, b in zip(r0["grads"], r1["grads"]))
