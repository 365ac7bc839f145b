"""Which (out-channel tiles per workgroup, tile groups per workgroup) the Winograd conv should take per layer shape: every shape of
a level's D+G step timed under the MG_WINO_CFG / MG_WINO_WT overrides against the dispatch default.  python tools/tune_wino.py [level] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops

level = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)

def timeit(fn, iters=15):
    for _ in range(8): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

tail = [128, 112, 96, 80, 64, 48, 32, 16]
ins = [32] + tail[:-1]
dch = [(16, 32), (32, 48), (48, 64), (64, 80), (80, 96), (96, 112), (112, 128), (128, 144), (144, 160)]
shapes = set()
s = 2
for i in range(level + 1):  # generator: conv ci->ci @s (N), dgrad same; conv ci->co @2s handled by the sub-pixel kernels
    shapes.add((N, ins[i], ins[i], s)); s *= 2
side = s
t = side
for i in range(7 - level, 9):  # critic: fwd / dgrad over 3N and N, tangent over N
    ci, co = dch[i]
    for n in (3 * N, N):
        shapes.add((n, ci, co, t)); shapes.add((n, co, ci, t))      # forward, data gradient (channels swapped)
        shapes.add((n, co, co, t // 2))
    t //= 2
total_def = total_best = 0.0
for (n, ci, co, h) in sorted(shapes, key=lambda q: -q[0] * q[1] * q[2] * q[3] * q[3]):
    if not ops.wino3x3_supported(n, co, h, h, cin=ci) or h < 8:
        continue
    x = R(n, ci, h, h); w = R(co, ci, 3, 3) * 0.05; b = R(co)
    up = ops.pack_wino3x3(w, dgrad=False)
    fn = lambda: ops.conv3x3(x, None, b, co, lrelu=True, wino=up)
    for k in ("MG_WINO_CFG", "MG_WINO_WT"): os.environ.pop(k, None)
    base = timeit(fn)
    res = {}
    for cfg in (2, 3, 4):
        for wt in (2, 4):
            os.environ["MG_WINO_CFG"], os.environ["MG_WINO_WT"] = str(cfg), str(wt)
            try:
                res[(cfg, wt)] = timeit(fn)
            except Exception as e:  # noqa: BLE001
                res[(cfg, wt)] = float("inf")
    for k in ("MG_WINO_CFG", "MG_WINO_WT"): os.environ.pop(k, None)
    best = min(res, key=res.get)
    total_def += base; total_best += min(base, res[best])
    flag = "" if res[best] > 0.97 * base else "  <== "
    print(f"n={n:4d} {ci:3d}->{co:3d} @{h:3d}: default {base:7.1f} us | best cfg{best[0]} wt{best[1]} {res[best]:7.1f} us | " +
          " ".join(f"{c}{w}:{res[(c, w)]:.0f}" for (c, w) in sorted(res)) + flag, flush=True)
print(f"sum over shapes (one launch each): default {total_def:.0f} us, best {total_best:.0f} us")
