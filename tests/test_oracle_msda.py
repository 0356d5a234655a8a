"""CPU: the two independent restatements of multi-scale deformable attention in oracle/msda.py (grid_sample formulation
vs explicit loops with the CUDA kernel's corner rule) agree, including samples outside the maps and exactly on borders."""
import torch

from oracle import msda as OM


def _case(seed=0, B=2, Q=5, NH=2, D=4, shapes=((3, 4), (2, 2)), P=3, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    S = sum(h * w for h, w in shapes)
    value = torch.randn((B, S, NH, D), generator=g, dtype=dtype)
    loc = torch.rand((B, Q, NH, len(shapes), P, 2), generator=g, dtype=dtype) * 1.6 - 0.3      # 30 % outside [0, 1]
    loc[0, 0, 0, 0, 0] = torch.tensor([0.0, 0.0]); loc[0, 0, 0, 0, 1] = torch.tensor([1.0, 1.0])
    loc[0, 1, 0, 0, 0] = torch.tensor([0.125, 0.5])       # exactly on a pixel centre (x = 0.125 * 4 - 0.5 = 0)
    att = torch.softmax(torch.randn((B, Q, NH, len(shapes) * P), generator=g, dtype=dtype), -1).view(B, Q, NH, len(shapes), P)
    return value, list(shapes), loc, att


def test_grid_sample_formulation_equals_naive_loops():
    value, shapes, loc, att = _case()
    a = OM.msda_grid_sample(value, shapes, loc, att)
    b = OM.msda_naive(value, shapes, loc, att)
    assert a.shape == (2, 5, 8) and float((a - b).abs().max()) < 1e-12
    assert float(a.abs().max()) > 0.1
