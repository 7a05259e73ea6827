import torch, time
dev = torch.device("cuda")
n_idx = 53_000_000
for gb in (2, 8, 16, 32, 50):
    n = gb * (1 << 30) // 8
    x = torch.empty(n, dtype=torch.int64, device=dev)
    idx = torch.randint(0, n, (n_idx,), device=dev)
    sidx, _ = torch.sort(idx)
    for name, ii in (("random", idx), ("sorted", sidx)):
        torch.cuda.synchronize(); y = x[ii]; torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): y = x[ii]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"table {gb} GB {name}: {dt*1e3:.2f} ms for {n_idx/1e6:.0f}M gathers -> {n_idx/dt/1e9:.1f} G/s", flush=True)
    t0 = time.perf_counter(); s2, _ = torch.sort(idx); torch.cuda.synchronize(); print(f"  sort of indices: {(time.perf_counter()-t0)*1e3:.2f} ms")
    del x, idx, sidx
