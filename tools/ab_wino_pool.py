"""A/B of mg_wino3x3 in its pooled-output-only epilogues (forward + tile mask out; tangent with mask bytes) between two builds of the
library (old.so new.so): interleaved timing + bitwise comparison.  python tools/ab_wino_pool.py OLD.so NEW.so"""
import ctypes, sys, torch
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
LRELU, MASK_AUX, POOL, MASK_OUT, MASK_BYTES = 2, 4, 16, 32, 64
libs = []
for name in sys.argv[1:3]:
    lib = ctypes.CDLL(name)
    lib.mg_wino3x3_packed_floats.restype = ctypes.c_size_t
    lib.mg_wino3x3_packed_floats.argtypes = [ctypes.c_int] * 2
    lib.mg_wino3x3_pack.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    f = lib.mg_wino3x3; f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_void_p]
    libs.append((name.split("/")[-1], lib))
cases = [(192, 48, 64, 128), (192, 64, 80, 64), (192, 80, 96, 32), (64, 48, 64, 128), (96, 64, 80, 64), (18, 32, 48, 256), (18, 16, 32, 512)]
for (n, ci, co, h) in cases:
    x = torch.randn(n, ci, h, h, device=dev, generator=g)
    w = torch.randn(co, ci, 3, 3, device=dev, generator=g) * 0.05
    b = torch.randn(co, device=dev, generator=g)
    res = []
    s = torch.cuda.current_stream().cuda_stream
    for name, lib in libs:
        up = torch.empty(lib.mg_wino3x3_packed_floats(ci, co), device=dev)
        lib.mg_wino3x3_pack(w.data_ptr(), up.data_ptr(), co, ci, 0, s)
        m = torch.empty(n, co, h // 2, h // 2, dtype=torch.uint8, device=dev)
        p = torch.empty(n, co, h // 2, h // 2, device=dev)
        p2 = torch.empty_like(p)
        fwd = (lambda lib=lib, up=up, m=m, p=p: lib.mg_wino3x3(x.data_ptr(), up.data_ptr(), b.data_ptr(), None, m.data_ptr(), p.data_ptr(), None,
                                                               n, ci, co, h, h, LRELU | POOL | MASK_OUT, 0.2, s))
        tan = (lambda lib=lib, up=up, m=m, p2=p2: lib.mg_wino3x3(x.data_ptr(), up.data_ptr(), None, m.data_ptr(), None, p2.data_ptr(), None,
                                                                 n, ci, co, h, h, MASK_AUX | POOL | MASK_BYTES, 0.2, s))
        assert fwd() == 0 and tan() == 0
        res.append((fwd, tan, m, p, p2))
    for which, idx in (("fwd+mask_out", 0), ("tangent bytes", 1)):
        times = [0.0, 0.0]
        for rep in range(3):
            for i, r in enumerate(res):
                for _ in range(10): r[idx]()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30): r[idx]()
                e1.record(); e1.synchronize()
                times[i] += e0.elapsed_time(e1) / 30 / 3
        same = all(torch.equal(res[0][k], res[1][k]) for k in (2, 3, 4))
        print(f"n={n:4d} {ci:3d}->{co:3d} @{h:3d} {which:14s}: {libs[0][0]} {times[0]*1e3:8.1f} us   {libs[1][0]} {times[1]*1e3:8.1f} us   ({times[1]/times[0]:.3f}x)  bitwise equal: {same}", flush=True)
