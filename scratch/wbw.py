import torch, time
x = torch.empty(134*1024*1024//4, device="cuda")
y = torch.empty_like(x)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
dt=t(lambda: x.zero_()); print("fill  GB/s", x.numel()*4/dt/1e9)
dt=t(lambda: y.copy_(x)); print("copy  GB/s (r+w)", 2*x.numel()*4/dt/1e9)
dt=t(lambda: x.sum()); print("read  GB/s", x.numel()*4/dt/1e9)
