"""Re-packing of every trainable layer's weight-derived forms in ONE launch per kernel family, right after the optimizer step.

The Winograd 3x3 kernels read ``G g G^T`` of every filter in their consumption order (wino.packed_pair) and the bf16x6 1x1 GEMMs the
three bf16 planes of the filter (gemm_bf6.pack_both); both are re-derived whenever the weight's version moves, i.e. once per training
step and layer, lazily at the layer's first use: 38 + 34 launches of 6-8 us per step in the BEVDepth4D recipe (0.54 ms of
launch-latency kernels between the step's convolutions).  ``WeightPacker.repack()`` does the same work for all layers in two launches
(dbev_wino_filter_pack_multi / dbev_gemm_bf16x6_pack_multi) into the SAME buffers the layers already hold, and re-keys the per-weight
caches so that the layers find their packs fresh.  The reference has no counterpart (cuDNN consumes ``nn.Conv2d.weight`` directly,
mmdet3d/models/bricks/res_block.py:102-230).

A layer takes part once it has run (its cache entry records the input size its pack format depends on); layers whose entry is already
fresh, or whose weight has not been packed yet, are skipped -- the lazy path stays the fallback for everything, and it writes a stale
entry's buffers again instead of replacing them (wino.packed_pair, gemm_bf6._cache_for), so a pack buffer keeps its address for the
life of its weight's storage either way.  `repack()` returns the weights it left STALE to the lazy path (`self.skipped`; empty in the
bench recipe, asserted by tests/test_gpu_graphed.py): a captured hipGraph re-validates what it baked in before every replay
(graphed.GraphedNoGrad), which is where such a layer is refreshed.
"""
import os

import numpy as np
import torch

from . import _lib as L

_ON = os.environ.get("DBEV_MULTI_PACK", "1") != "0"

# dbevPackJob of include/dbev_hip.h
_JOB = np.dtype([("weight", np.uint64), ("so", np.int64), ("sc", np.int64), ("sa", np.int64), ("sb", np.int64),
                 ("Cout", np.int32), ("Cin", np.int32), ("kind_a", np.int32), ("kind_b", np.int32),
                 ("out_a", np.uint64), ("out_b", np.uint64)], align=True)
assert _JOB.itemsize == 72


class WeightPacker:
    def __init__(self, roots):
        from .gemm_bf6 import Bf6Conv2d, Bf6Conv3x3S2, Bf6Linear
        from .wino import WinoConv2d
        mods = [m for r in roots if r is not None for m in r.modules()]
        self.wino = [m.weight for m in mods if type(m) is WinoConv2d and m.weight.requires_grad]
        self.bf6 = [m.weight for m in mods if type(m) in (Bf6Conv2d, Bf6Conv3x3S2, Bf6Linear) and m.weight.requires_grad]
        self._tables = {}                      # family -> (signature, device table, n jobs, max size, per-job (weight, new cache entry maker))
        self.launches = 0
        self.skipped = []                      # of the last repack(): stale weights left to the lazy path, with the reason

    # ---- one family ------------------------------------------------------------------------------------------------------------
    def _wino_jobs(self):
        rows, fix = [], []
        for w in self.wino:
            hit = getattr(w, "_dbev_wino_pair", None)
            if hit is None:
                continue                                        # never packed: the layer has not run yet
            if hit[0][1] != w.data_ptr():
                self.skipped.append((w, "wino: the weight's storage moved"))
                continue
            (ver, ptr, N, H, W), fwd, dgrad = hit[0], hit[1], hit[2]
            if ver == w._version:
                continue                                        # already fresh
            Co, C = int(w.shape[0]), int(w.shape[1])
            fk = int(L.call("dbev_wino_conv3x3_forward_kernel", N, H, W, C, Co))
            dk = int(L.call("dbev_wino_conv3x3_forward_kernel", N, H, W, Co, C)) if dgrad is not None else 0
            if fk == 0:
                self.skipped.append((w, "wino: no forward kernel for the recorded input size"))
                continue
            so, sc, sa, sb = w.stride()
            rows.append((w.data_ptr(), so, sc, sa, sb, Co, C, fk, dk, fwd.data_ptr(), 0 if dgrad is None else dgrad.data_ptr()))
            fix.append((w, "_dbev_wino_pair", lambda w=w, N=N, H=H, W=W, fwd=fwd, dgrad=dgrad: ((w._version, w.data_ptr(), N, H, W), fwd, dgrad,
                                                                                              L.fingerprint(w))))
        return rows, fix, max((r[5] * r[6] for r in rows), default=0)

    def _bf6_jobs(self):
        rows, fix = [], []
        for w in self.bf6:
            hit = getattr(w, "_dbev_bf6_packs", None)
            if hit is None or hit[0][0] == w._version and hit[0][1] == w.data_ptr():
                continue                                        # never packed / already fresh
            if hit[0][1] != w.data_ptr():
                self.skipped.append((w, "bf6: the weight's storage moved"))
                continue
            packs, spare = hit[1], (hit[3] if len(hit) > 3 else {})
            from .gemm_bf6 import matrix
            w2 = matrix(w)                                      # [Cout, K]: the 1x1 filter, or the channels-last memory of a 3x3 one
            if w2.data_ptr() != w.data_ptr():
                self.skipped.append((w, "bf6: a k x k filter that is not channels-last (no matrix view)"))
                continue
            Co, Ci = int(w2.shape[0]), int(w2.shape[1])
            fwd = [(tn, b) for (tr, tn), b in packs.items() if not tr]
            dgr = [(tn, b) for (tr, tn), b in packs.items() if tr]
            if len(fwd) > 1 or len(dgr) > 1 or not (fwd or dgr):
                self.skipped.append((w, "bf6: several tile widths in use for one weight" if (fwd or dgr) else "bf6: an entry without packs"))
                continue
            ka, oa = (fwd[0][0], fwd[0][1].data_ptr()) if fwd else (0, 0)
            kb, ob = (dgr[0][0], dgr[0][1].data_ptr()) if dgr else (0, 0)
            rows.append((w.data_ptr(), w2.stride(0), w2.stride(1), 0, 0, Co, Ci, ka, kb, oa, ob))
            fix.append((w, "_dbev_bf6_packs", lambda w=w, packs=packs, spare=spare: ((w._version, w.data_ptr()), packs, L.fingerprint(w), spare)))
        return rows, fix, max((r[5] * r[6] // 8 for r in rows), default=0)

    def _run(self, family, entry, rows, fix, size, dev):
        if not rows:
            return
        sig = tuple(rows)
        kept = self._tables.get(family)
        if kept is None or kept[0] != sig:                      # the table only changes when a buffer or a shape does: built and uploaded once
            host = np.zeros((len(rows),), dtype=_JOB)
            for i, r in enumerate(rows):
                host[i] = r
            table = L.h2d(torch.from_numpy(host.view(np.uint8).copy()), dev)
            kept = (sig, table)
            self._tables[family] = kept
        with torch.cuda.device(dev):
            L.call(entry, L.ptr(kept[1]), len(rows), int(size), L.stream_ptr(dev))
        self.launches += 1
        for w, attr, make in fix:
            setattr(w, attr, make())

    @torch.no_grad()
    def repack(self):
        """call after the optimizer step (the version counters of the updated weights have moved) -> the stale weights it did NOT
        re-pack (left to the lazy path, which refreshes their buffers in place at the layer's next use)"""
        self.skipped = []
        if not _ON:
            return self.skipped
        ws = self.wino or self.bf6
        if not ws or not ws[0].is_cuda:
            return self.skipped
        dev = ws[0].device
        rows, fix, size = self._wino_jobs()
        self._run("wino", "dbev_wino_filter_pack_multi", rows, fix, size, dev)
        rows, fix, size = self._bf6_jobs()
        self._run("bf6", "dbev_gemm_bf16x6_pack_multi", rows, fix, size, dev)
        return self.skipped
