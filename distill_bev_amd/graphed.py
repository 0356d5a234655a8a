"""HIP-graph replay of a gradient-free, fixed-shape piece of the training step.

The adjacent camera frame of BEVDepth4D goes through the image backbone and neck without a gradient (bevdet_distill_more.py:389-441:
its BEV map is detached; detectors.extract_img_feat runs it under no_grad): ~330 kernel launches whose host side (module calls,
autograd.Function.apply, ctypes marshalling) is a pure replayable sequence -- same shapes, same buffers, no host decision depends on
device data.  `GraphedNoGrad(fn)` captures `fn(x)` into a hipGraph the third time it sees an input signature and replays it
afterwards: the input is copied into the graph's static input, the output is the graph's static output (valid until the next
replay).  Measured (tools/fwd_host_vs_gpu.py, tools/ab_env.sh DBEV_GRAPH_ADJ 0 1): the host issues the step's forward in 23.7 ->
18 ms against 48 ms of GPU time -- the step is NOT host-bound on an idle host, the replay buys 0.2-0.6 ms per step (shorter bubbles
behind the forward's one synchronising call) and headroom when eight ranks share one host's cores.

What makes a replay equal to an eager call:
  * the kernels read the layers' derived weight forms (Winograd filter packs, bf16 planes, folded conv + norm packs, eval-mode norm
    coefficients) through raw addresses.  Every accessor of such a buffer reports what it hands out while the capture runs
    (_lib.DERIVED_LOG); the graph entry keeps those tensors ALIVE (a replay can never read freed memory) and, before every replay,
    calls each accessor again (_lib.revalidate): an entry whose source moved on (optimizer step, load_state_dict, any write that
    bumps the version counter) is re-derived by the accessor INTO THE SAME BUFFER -- stream-ordered before the replay -- and an
    accessor that hands out a different buffer than the one baked in (a cache dropped by invalidate_eval_coef, a new input size,
    a weight on new storage) drops the graph: the call runs eagerly and the capture is redone after a fresh warm-up.
    packer.WeightPacker.repack (one launch per family after the optimizer step, into the same buffers) makes the re-validation a
    pure host-side check in the training loop; it is an optimisation, not what correctness rests on (DBEV_MULTI_PACK=0 and a packer
    that skips layers are covered by tests/test_gpu_graphed.py);
  * training-mode norms update their running statistics inside the captured kernels (in place, stable addresses);
  * `valid_token()` -- anything that changes module state behind the graph's back (train()/eval(), a different parameter / buffer
    tensor) changes the token; the graph is dropped and captured again after `warmup` eager calls in the new state.
The kernel event log (dbev_kernel_timing_*) is host code: replayed launches are not logged.  Nothing is captured while any timing is
on, and while EVERY kernel is logged (mask -1, or the per-entry-point brackets: bench.py's instrumented steps) the call runs eagerly.
detectors.extract_img_feat uses the graph in TRAINING mode only (validation runs eagerly)."""
import os

import torch

_ON = os.environ.get("DBEV_GRAPH_ADJ", "1") != "0"
_FORCED = os.environ.get("DBEV_GRAPH_ADJ") == "1"


def enabled(world_size=1):
    """DBEV_GRAPH_ADJ=0: never; =1: always; unset: in single-process runs only -- capture beside a live RCCL communicator (its watchdog
    thread) is covered by a two-rank test on one GPU (gloo) but has not been run on a multi-GPU node, and a failed capture is fatal"""
    return _ON and (_FORCED or world_size == 1)


class GraphedNoGrad:
    def __init__(self, fn, token=None, warmup=2, norms=None):
        """norms: callable -> the norm modules inside `fn`: the captured kernels update the running statistics of those in TRAINING
        mode in place WITHOUT the version bump the eager path gives them (_lib.touched), so the eval-mode coefficients kept per module
        keyed on those versions (bn_act._eval_coef) are dropped after every replay -- of the training-mode norms only: an eval-mode
        norm inside the graph has its coefficient tensor baked in.  (Bumping the versions here instead would trip autograd's
        saved-tensor check of a stock batch_norm that saved the same buffers in the key frame's forward.)"""
        self.fn, self.token, self.warmup, self.norms = fn, token, warmup, norms
        self.graphs = {}
        self.replays = self.captures = self.eager = self.dropped = 0
        self.failed = None             # the exception of a capture that did not work

    def reset(self):
        self.graphs.clear()

    def _eager(self, x):
        self.eager += 1
        with torch.no_grad():
            return self.fn(x)

    def _drop(self, key):
        """forget the captured graph of `key` and start its warm-up again (the next `warmup` calls run eagerly)"""
        if self.graphs.pop(key, None) is not None:
            self.dropped += 1
        self.graphs.pop(("seen",) + key, None)

    @staticmethod
    def _deps_fresh(ent):
        """re-validate every derived buffer the capture read (see the module docstring) -> False when one of them moved"""
        from . import _lib as L
        for kind, owner, args, ptrs in ent["deps"]:
            now = L.revalidate(kind, owner, args)
            if tuple(0 if t is None else t.data_ptr() for t in now) != ptrs:
                return False
        return True

    def __call__(self, x):
        from . import _lib as L
        if (not (_ON and x.is_cuda) or self.failed is not None or L.CHECK_PACKS or torch.cuda.is_current_stream_capturing()
                or L.kernel_timing_active(full=True)):
            # (DBEV_CHECK_PACKS reads fingerprints back to the host: not capturable; every kernel being logged: bench.py's instrumented
            # steps see all launches)
            return self._eager(x)
        tok = self.token() if self.token is not None else None
        key = (tuple(x.shape), x.dtype, x.device)
        ent = self.graphs.get(key)
        if ent is not None and ent["token"] != tok:
            self._drop(key)                       # module state changed: the new state gets its own eager warm-up calls
            ent = None
        if ent is not None and not self._deps_fresh(ent):
            self._drop(key)                       # a baked-in buffer is no longer the layer's: eager now, captured again later
            ent = None
        if ent is None:
            seen = self.graphs.setdefault(("seen",) + key, {"n": 0, "token": tok})
            if seen["token"] != tok:
                seen["n"], seen["token"] = 0, tok
            seen["n"] += 1
            if seen["n"] <= self.warmup or L.kernel_timing_active():
                return self._eager(x)             # the first calls run eagerly: library plans, MIOpen solutions, lazy packs settle
            try:
                ent = self._capture(x, tok)
            except Exception as e:                # a failed capture leaves the stream invalidated on this stack: no quiet way back
                self.failed = e
                raise RuntimeError("distill_bev_amd.graphed: hipGraph capture of the gradient-free frame failed "
                                   f"({type(e).__name__}: {e}); run with DBEV_GRAPH_ADJ=0") from e
            self.graphs[key] = ent
        ent["x"].copy_(x)
        ent["graph"].replay()
        self.replays += 1
        if self.norms is not None:
            for m in self.norms():
                if m.training:
                    m.__dict__.pop("_dbev_eval_coef", None)
        return ent["y"]

    def _capture(self, x, tok):
        from . import _lib as L
        static_x = torch.empty_like(x)
        static_x.copy_(x)
        torch.cuda.synchronize(x.device)
        graph = torch.cuda.CUDAGraph()
        log, L.DERIVED_LOG = L.DERIVED_LOG, []
        try:
            with torch.no_grad():
                # thread_local: what other threads do meanwhile (a collective's watchdog, a data loader) does not invalidate the capture
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    y = self.fn(static_x)
            noted = L.DERIVED_LOG
        finally:
            L.DERIVED_LOG = log
        torch.cuda.synchronize(x.device)
        self.captures += 1
        deps, keep, seen = [], [], set()
        for kind, owner, args, tensors in noted:
            ptrs = tuple(0 if t is None else t.data_ptr() for t in tensors)
            sig = (kind, id(owner), repr(args) if kind != "wino_folded" else (id(args[0]),) + tuple(args[1:]), ptrs)
            if sig in seen:
                continue
            seen.add(sig)
            deps.append((kind, owner, args, ptrs))
            keep.append(tensors)                  # alive as long as the graph: a replay never reads freed memory
        return {"graph": graph, "x": static_x, "y": y, "token": tok, "deps": deps, "keep": keep}

    def baked_in(self):
        """[(kind, owner, buffer addresses)] of every captured graph (tests, diagnostics)"""
        return [(k, o, p) for key, ent in self.graphs.items() if key[0] != "seen" for k, o, _a, p in ent["deps"]]


def state_token(*modules):
    """changes when something a captured graph baked in may have changed: the identity of a parameter / buffer tensor or its storage, a
    module's training flag.  (Values may change freely -- the graph reads them through the same addresses.)"""
    t = []
    for m in modules:
        if m is None:
            continue
        for mod in m.modules():
            t.append(mod.training)
        for p in m.parameters():
            t.append(p.data_ptr())
        for b in m.buffers():
            t.append(b.data_ptr())
    return hash(tuple(t))
