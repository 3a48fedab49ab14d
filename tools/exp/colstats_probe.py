"""Developer probe: column statistics of a tall [V, d] matrix on ROCm -- torch.var_mean(dim=0) vs two-level."""
import torch, time
d = torch.device("cuda:0")
def two_level(x, c=2048):
    V, dd = x.shape; S = V // c
    v_s, m_s = torch.var_mean(x[:S * c].view(S, c, dd), dim=1, unbiased=False)
    r = V - S * c
    tot = m_s.sum(0) * c
    if r: 
        v_r, m_r = torch.var_mean(x[S * c:], dim=0, unbiased=False); tot = tot + m_r * r
    mean = tot / V
    acc = c * (v_s + (m_s - mean) ** 2).sum(0)
    if r: acc = acc + r * (v_r + (m_r - mean) ** 2)
    return acc / V, mean
def bench(f, *a):
    for _ in range(2): f(*a)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): f(*a)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / 5 * 1e3
for dd in (60, 71):
    x = torch.randn(4_594_386, dd, device=d) * 2 + 1
    ref = torch.var_mean(x.double(), dim=0, unbiased=False)
    for name, f in (("var_mean dim0", lambda x: torch.var_mean(x, dim=0, unbiased=False)), ("two-level", two_level),
                    ("two-level c=512", lambda x: two_level(x, 512)), ("two-level c=8192", lambda x: two_level(x, 8192))):
        v, m = f(x)
        print(dd, name, f"{bench(f, x):.3f} ms", "err var", float((v - ref[0]).abs().max()), "mean", float((m - ref[1]).abs().max()))
