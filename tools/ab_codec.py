"""Bitwise A/B of mg_codec_fwd between two builds of the library (old.so new.so) on a synthetic 10-minute STFT, plus timing."""
import ctypes, sys, torch
dev = torch.device("cuda", 0)
T = 103360
g = torch.Generator(device=dev).manual_seed(5)
c = torch.view_as_real(torch.randn(512, T, dtype=torch.complex64, device=dev)).contiguous()
bark = torch.rand(512, device=dev) + 0.5
outs = []
for name in sys.argv[1:]:
    lib = ctypes.CDLL(name)
    lib.mg_codec_fwd_ws_bytes.restype = ctypes.c_size_t; lib.mg_codec_fwd_ws_bytes.argtypes = [ctypes.c_int]
    f = lib.mg_codec_fwd; f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    ws = torch.empty(lib.mg_codec_fwd_ws_bytes(T), dtype=torch.uint8, device=dev)
    S = (T - 1) // 512
    m = torch.empty(S, 512, 512, device=dev); p = torch.empty(S, 512, 512, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: f(c.data_ptr(), bark.data_ptr(), m.data_ptr(), p.data_ptr(), ws.data_ptr(), ws.numel(), T, 512, s)
    rc = run()
    if rc != 0:
        lib.mg_last_error.restype = ctypes.c_char_p
        raise SystemExit(f"{name}: rc={rc}: {lib.mg_last_error().decode()}")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); e1.synchronize()
    print(name.split("/")[-1], f"{e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
    outs.append((m.clone(), p.clone()))
if len(outs) == 2:
    print("magn bitwise equal:", torch.equal(outs[0][0], outs[1][0]), " phase bitwise equal:", torch.equal(outs[0][1], outs[1][1]))
