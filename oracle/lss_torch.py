"""Oracle (CPU, torch) restatement of the reference's *own CPU execution path* of the
splat, used as the timed ``cpu_baseline`` ("port") in bench.py and cross-checked
against oracle/lss.py in tests.  TEST / BASELINE INFRASTRUCTURE -- never imported by
the product package.

Follows mmdet3d/models/necks/view_transformer_mine.py:141-181 (voxel_pooling:
truncating index, range mask, rank, argsort, gather, cumsum trick :30-47, scatter
into the (B, C, Z, Y, X) grid, Z collapse) op for op with torch CPU tensors, so the
time measured is what the reference's CPU path costs on the same host cores.
"""
import torch


class _CumsumTrick(torch.autograd.Function):
    """view_transformer_mine.py:30-56 (QuickCumsum): forward = cumsum, keep the last row of
    every rank run, adjacent difference; backward = each point receives the gradient of its run
    (a gather through cumsum(kept) - 1), NOT autograd through the cumsum."""

    @staticmethod
    def forward(ctx, feats, rank):
        csum = feats.cumsum(0)
        last = torch.ones(csum.shape[0], dtype=torch.bool, device=feats.device)
        last[:-1] = rank[1:] != rank[:-1]
        csum = csum[last]
        ctx.save_for_backward(last)
        return torch.cat((csum[:1], csum[1:] - csum[:-1])), last

    @staticmethod
    def backward(ctx, gsum, _glast):
        (last,) = ctx.saved_tensors
        back = torch.cumsum(last, 0)
        back[last] -= 1
        return gsum[back], None


def voxel_pooling_cumsum(geom, x, dx, bx, nx):
    """geom f32[B,N,D,H,W,3], x f32[B,N,D,H,W,C] (torch CPU) -> f32[B, C*Z, Y, X]."""
    B, N, D, H, W, C = x.shape
    n_pts = B * N * D * H * W
    nxl = nx.to(torch.long)
    feats = x.reshape(n_pts, C)
    cell = ((geom - (bx - dx / 2.0)) / dx).long().view(n_pts, 3)          # :150 trunc
    batch = torch.arange(B, dtype=torch.long, device=x.device).repeat_interleave(n_pts // B).view(-1, 1)
    cell = torch.cat((cell, batch), 1)
    ok = ((cell[:, 0] >= 0) & (cell[:, 0] < nxl[0]) & (cell[:, 1] >= 0) & (cell[:, 1] < nxl[1])
          & (cell[:, 2] >= 0) & (cell[:, 2] < nxl[2]))                      # :157-159
    feats, cell = feats[ok], cell[ok]
    rank = (cell[:, 0] * (nxl[1] * nxl[2] * B) + cell[:, 1] * (nxl[2] * B)
            + cell[:, 2] * B + cell[:, 3])                                  # :164-167
    order = rank.argsort()
    feats, cell, rank = feats[order], cell[order], rank[order]
    sums, last = _CumsumTrick.apply(feats, rank)                            # :172 QuickCumsum.apply
    cell = cell[last]
    grid = torch.zeros((B, C, int(nxl[2]), int(nxl[1]), int(nxl[0])), device=x.device)
    grid[cell[:, 3], :, cell[:, 2], cell[:, 1], cell[:, 0]] = sums          # :176
    return torch.cat(grid.unbind(dim=2), 1)                                 # :178


def lift(depth_prob, img_feat, B, N):
    """bevdet_distill_more.py:413-416: volume = depth[:,None] * feat[:,:,None] ->
    [B,N,D,H,W,C] (materialised, as the reference does)."""
    BN, D, H, W = depth_prob.shape
    C = img_feat.shape[1]
    vol = depth_prob.unsqueeze(1) * img_feat.unsqueeze(2)
    return vol.view(B, N, C, D, H, W).permute(0, 1, 3, 4, 5, 2)
