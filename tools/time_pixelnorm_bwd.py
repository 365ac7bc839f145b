import os, sys, torch
sys.path.insert(0, '/root/repo')
from musicgan_amd import ops
dev = torch.device("cuda", 0)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters
for (n, c, s) in [(64, 48, 128), (64, 64, 64), (64, 80, 32), (64, 96, 16), (64, 112, 8)]:
    gp = torch.randn(n, c, s, s, device=dev); p = torch.randn(n, c, s, s, device=dev); rn = torch.rand(n, 1, s, s, device=dev) + 0.5
    os.environ.pop("MG_PN_BWD_NOLDS", None)
    a = timeit(lambda: ops.pixelnorm_lrelu_bwd(gp, p, rn, from_p=True))
    os.environ["MG_PN_BWD_NOLDS"] = "1"
    b = timeit(lambda: ops.pixelnorm_lrelu_bwd(gp, p, rn, from_p=True))
    gb = 4 * gp.numel() * 4 / 1e9
    print(f"N={n} C={c} {s}x{s}: lds {a*1e3:.1f} us ({gb/a*1e3:.0f} GB/s alg 4-pass)  two-pass {b*1e3:.1f} us")
