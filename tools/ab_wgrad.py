"""A/B of mg_wino3x3_wgrad between two builds of the library (old.so new.so): interleaved timing + bitwise comparison."""
import ctypes, sys, torch
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
libs = []
for name in sys.argv[1:3]:
    lib = ctypes.CDLL(name)
    lib.mg_wino3x3_wgrad_ws_bytes.restype = ctypes.c_size_t
    lib.mg_wino3x3_wgrad_ws_bytes.argtypes = [ctypes.c_int] * 5
    f = lib.mg_wino3x3_wgrad; f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_size_t] + [ctypes.c_int] * 8 + [ctypes.c_void_p]
    libs.append((name.split("/")[-1], lib))
cases = [(192, 48, 64, 128, 0), (192, 64, 80, 64, 0), (192, 64, 64, 64, 0), (192, 80, 96, 32, 0), (64, 64, 48, 128, 1), (64, 80, 64, 64, 1),
         (192, 96, 112, 16, 0), (18, 16, 32, 512, 0), (18, 32, 48, 256, 0), (6, 32, 16, 512, 1), (32 * 3, 64, 80, 64, 0)]
for (n, ci, co, h, ups) in cases:
    x = torch.randn(n, ci, h // 2 if ups else h, h // 2 if ups else h, device=dev, generator=g)
    gy = torch.randn(n, co, h, h, device=dev, generator=g)
    res, outs = [], []
    for name, lib in libs:
        ws = torch.empty(lib.mg_wino3x3_wgrad_ws_bytes(n, ci, co, h, h), dtype=torch.uint8, device=dev)
        gw = torch.empty(co, ci, 3, 3, device=dev); gb = torch.empty(co, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        run = (lambda lib=lib, ws=ws, gw=gw, gb=gb: lib.mg_wino3x3_wgrad(x.data_ptr(), gy.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(),
                                                                         ws.numel(), n, ci, co, h, h, 1 if ups else 0, 0, 0, s))
        assert run() == 0
        res.append((run, gw, gb))
    times = [0.0, 0.0]
    for rep in range(3):  # interleaved
        for i, (run, gw, gb) in enumerate(res):
            for _ in range(5): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); e1.synchronize()
            times[i] += e0.elapsed_time(e1) / 10 / 3
    same = torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    print(f"n={n:4d} {ci:3d}x{co:3d} @{h:3d} ups={ups}: {libs[0][0]} {times[0]*1e3:8.1f} us   {libs[1][0]} {times[1]*1e3:8.1f} us   ({times[1]/times[0]:.3f}x)  bitwise equal: {same}", flush=True)
