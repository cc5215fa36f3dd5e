import torch, time
x = torch.empty(14041824, dtype=torch.float64, device="cuda")
y = torch.empty(6_000_000, dtype=torch.float64, device="cuda")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best * 1e3
print("fill 112MB us", t(lambda: x.fill_(1.5)))
print("read 48MB (sum) us", t(lambda: y.sum()))
z = torch.empty_like(x)
print("copy 112MB us", t(lambda: z.copy_(x)))
print("empty kernel us", t(lambda: y[:1].fill_(0.0)))
