#!/usr/bin/env python3
"""How exact is an fp32 GEMM assembled from bf16 products?  (CPU, numpy; dev tool behind the "bf16 matrix cores for fp32 work" note of
DESIGN.md)  x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (every difference exact in fp32); products
of bf16 values are exact in fp32, sums are accumulated in fp32.  Compared with a plain fp32 GEMM against an fp64 reference."""
import numpy as np
rng = np.random.default_rng(0)
def bf16(x, trunc=False):
    u = x.astype(np.float32).view(np.uint32)
    if trunc:
        return (u & np.uint32(0xffff0000)).view(np.float32)
    r = ((u >> 16) & 1) + np.uint32(0x7fff)
    return ((u + r) & np.uint32(0xffff0000)).view(np.float32)
def split(x, n, trunc):
    parts, r = [], x.astype(np.float32)
    for _ in range(n):
        p = bf16(r, trunc); parts.append(p); r = (r - p).astype(np.float32)
    return parts
def gemm32(a, b):
    return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float32)
for (M, K, N, relu) in [(256, 256, 256, False), (256, 256, 256, True), (256, 2304, 128, True), (256, 64, 256, True)]:
    a = rng.standard_normal((M, K)).astype(np.float32)
    if relu: a = np.maximum(a, 0)
    b = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    sc = np.abs(ref).max()
    out = {"fp32": gemm32(a, b)}
    for trunc in (False, True):
        A, B = split(a, 3, trunc), split(b, 3, trunc)
        for name, terms in (("x3", [(0, 0), (0, 1), (1, 0)]), ("x6", [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]),
                            ("x9", [(i, j) for i in range(3) for j in range(3)])):
            acc = np.zeros((M, N), np.float32)
            for (i, j) in sorted(terms, key=lambda t: -(t[0] + t[1])):     # small terms first
                acc = (acc + gemm32(A[i], B[j])).astype(np.float32)
            out[f"bf16{name}{'t' if trunc else 'r'}"] = acc
    print(f"M{M} K{K} N{N} relu={relu}: " + "  ".join(f"{k} {np.abs(v - ref).max() / sc:.2e}" for k, v in out.items()))
