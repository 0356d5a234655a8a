#!/usr/bin/env python3
"""Where do the small ATen launches of one bench-configuration step come from?  A TorchDispatchMode counts fill / zero / copy / add
dispatches by (op, element count class, innermost frame under the repo or mmdet-style module code).
usage: aten_sites.py [ops-regex] [workload]   (workload: a bench_workloads name, e.g. bevformer_distill; default: the distillation step).
Round 6: sites are also ranked by the ELEMENTS they move (count x numel): which glue is worth a kernel."""
import collections, os, re, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
from distill_bev_amd.train_step import Trainer, build_model, make_batch

pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else r"zero|fill|full|copy|clone|empty_strided|add|mul|sub|div|cat|sum|mean|contiguous")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counts = collections.Counter()
elems = collections.Counter()


def _numel(args):
    for a in args:
        if isinstance(a, torch.Tensor):
            return a.numel()
        if isinstance(a, (list, tuple)):
            for b in a:
                if isinstance(b, torch.Tensor):
                    return b.numel()
    return -1


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if pat.search(name):
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if fr.filename.startswith(ROOT) and "aten_sites" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                    break
            n = _numel(args)
            if site == "?":
                node = torch._C._current_autograd_node()
                site = "? node=" + (node.name() if node is not None else "-")
            if n < 0 and args and isinstance(args[0], (list, tuple)):
                n = int(np.prod(args[0])) if len(args[0]) else 1
            cls = "?" if n < 0 else ("<=4K" if n <= 4096 else "<=1M" if n <= (1 << 20) else ">1M")
            counts[(name, cls, site)] += 1
            elems[(name, site)] += max(n, 0)
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda:0")
if len(sys.argv) > 2:
    sys.path.insert(0, ROOT)
    import bench_workloads as BW
    wl = BW.WORKLOADS[sys.argv[2]](dev, 0, 1)
    step = wl.step
else:
    model, cfg = build_model(allow_synthetic_teacher=True)
    tr = Trainer(model, cfg, dev, channels_last=True)
    batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
    step = lambda: tr.step(batch)
for _ in range(3):
    step()
torch.cuda.synchronize()
with Sites():
    step()
torch.cuda.synchronize()
tot = collections.Counter()
for (name, cls, site), c in counts.items():
    tot[name] += c
print("dispatches per op:", dict(tot.most_common(40)))
for (name, cls, site), c in counts.most_common(120):
    print(f"{c:5d}  {name:38s} {cls:5s} {site}")
print("---- by elements moved (count x numel of the first tensor argument), M elements:")
for (name, site), e in elems.most_common(45):
    print(f"{e / 1e6:9.1f}  {name:38s} {site}")
