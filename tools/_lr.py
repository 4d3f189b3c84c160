import torch, time
dev = torch.device("cuda:0")
for n, m in ((60000, 100), (1000000, 50)):
    Z = torch.randn(n, m, device=dev); y = torch.randn(n, device=dev)
    def T(f, reps=3):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        return r, [round(t, 2) for t in ts]
    Zd, t = T(lambda: Z.double()); print(n, m, "Z.double()", t)
    G, t = T(lambda: Zd.t() @ Zd); print("  Zd^T Zd fp64", t)
    G32, t = T(lambda: Z.t() @ Z); print("  Z^T Z fp32", t)
    v = y.double()
    b, t = T(lambda: Zd.t() @ v); print("  Zd^T v", t)
    tt = torch.randn(m, device=dev, dtype=torch.float64)
    r, t = T(lambda: Zd @ tt); print("  Zd t", t)
    C = G + torch.eye(m, device=dev, dtype=torch.float64)
    L, t = T(lambda: torch.linalg.cholesky(C)); print("  cholesky", t)
