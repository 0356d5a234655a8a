import time, torch
x = torch.zeros(64*1024*1024, device="cuda")   # 256 MB: add_ ~ 100 us
torch.cuda.synchronize()
t0 = time.perf_counter(); stamps = []
for i in range(6000):
    x.add_(1.0)
    if i % 250 == 249:
        stamps.append((i + 1, (time.perf_counter() - t0) * 1e3))
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) * 1e3
print("total %.1f ms for 6000 launches (%.1f us each on the GPU)" % (tot, tot / 6))
print(" ".join("%d:%.0f" % s for s in stamps))
