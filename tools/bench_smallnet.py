"""Where the time of the fused small-map tail goes: the critic-tail forward program (conv2 of the 8x8 block .. classifier) timed as is
and in ablated builds of csrc/smallnet.hip (-DSN_NO_MFMA: filter stream without the matrix instructions; -DSN_NO_STREAM: matrix
instructions on one resident filter group).  python tools/bench_smallnet.py [n_images ...]"""
import ctypes, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import _lib, ops, _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda", 0)


def variant(defs):
    if not defs:
        return _lib.load()
    out = f"/tmp/sn_{'_'.join(defs)}.so"
    flags = [f for f in _build.FLAGS if not f.startswith("-Rpass")]
    cmd = [_build.HIPCC, *flags, *[f"-D{d}" for d in defs], "-shared", os.path.join(ROOT, "musicgan_amd/csrc/smallnet.hip"),
           os.path.join(ROOT, "musicgan_amd/csrc/core.hip"), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    lib = ctypes.CDLL(out)
    lib.mg_smallnet.restype = ctypes.c_int
    lib.mg_smallnet.argtypes = _lib.SIGNATURES["mg_smallnet"][1]
    return lib


def program(n, chans=(128, 144, 160)):
    g = torch.Generator(device=dev).manual_seed(1)
    c6, c7, c8 = chans
    R = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.05
    ws = [R(c6, c6, 3, 3), R(c7, c6, 3, 3), R(c7, c7, 3, 3), R(c8, c7, 3, 3), R(c8, c8, 3, 3)]
    wp = [ops.pack_smallnet(w, False) for w in ws]
    bs = [R(w.shape[0]) for w in ws]
    q = R(n, c6, 4, 4)
    E = lambda c, h: torch.empty(n, c, h, h, device=dev)
    sn = ops.SmallNet(1 if n <= 256 else 2)
    keep = [ws, wp, bs, q]
    sn.load(0, q)
    sn.conv(0, 1, wp[0], c6, c6, 4, 4, bias=bs[0], lrelu=True, out=E(c6, 4))
    sn.conv(1, 2, wp[1], c6, c7, 4, 4, bias=bs[1], lrelu=True, out=E(c7, 4))
    sn.pool(2, 0, c7, 4, 4, out=E(c7, 2))
    sn.conv(0, 1, wp[2], c7, c7, 2, 2, bias=bs[2], lrelu=True, out=E(c7, 2))
    sn.conv(1, 2, wp[3], c7, c8, 2, 2, bias=bs[3], lrelu=True, out=E(c8, 2))
    sn.pool(2, 0, c8, 2, 2, out=E(c8, 1))
    sn.conv(0, 1, wp[4], c8, c8, 1, 1, bias=bs[4], lrelu=True, out=E(c8, 1))
    sn.linear(1, c8, R(1, c8), R(1), torch.empty(n, 1, device=dev))
    keep.append(list(sn.keep))
    return sn, keep


def timeit(lib, sn, n, reps=200):
    arr = (_lib.SnOp * len(sn.ops))(*sn.ops)
    s = torch.cuda.current_stream().cuda_stream
    call = lambda: lib.mg_smallnet(ctypes.cast(arr, ctypes.c_void_p), len(sn.ops), n, sn.g, (int(sn.buf_floats) + 3) & ~3, 0.2,
                                   ctypes.c_void_p(s))
    for _ in range(20):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def cumulative(n):
    """Time of the first k ops of the program, k = 1 .. all (the differences are the ops' own times)."""
    sn, keep = program(n)
    lib = _lib.load()
    names = {v: k for k, v in vars(_lib).items() if k.startswith("MG_SN_") and isinstance(v, int) and k not in ("MG_SN_LRELU", "MG_SN_MASK_AUX", "MG_SN_NOLDS", "MG_SN_MAX_OPS")}
    full = list(sn.ops)
    prev = 0.0
    for k in range(1, len(full) + 1):
        sn.ops = full[:k]
        t = timeit(lib, sn, n)
        o = full[k - 1]
        print(f"   ops 1..{k:2d}  {t:6.1f} us  (+{t - prev:5.1f})  {names.get(o.op, o.op)} C {o.C} C2 {o.C2} {o.H}x{o.W}")
        prev = t


if __name__ == "__main__":
    if "--cumulative" in sys.argv:
        cumulative(24)
        sys.exit(0)
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8, 24, 96, 192]
    variants = [[], ["SN_NO_MFMA"], ["SN_NO_STREAM"]] + ([[d] for d in os.environ.get("SN_EXTRA", "").split(",") if d])
    libs = [(v, variant(v)) for v in variants]
    for n in sizes:
        sn, keep = program(n)
        print(f"n = {n:4d}: " + "   ".join(f"{'+'.join(v) or 'product'} {timeit(lib, sn, n):6.1f} us" for v, lib in libs), flush=True)
