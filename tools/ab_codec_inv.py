"""Bitwise A/B of mg_codec_inv between two builds of the library (old.so new.so), plus timing."""
import ctypes, sys, torch
dev = torch.device("cuda", 0)
N, W = 40, 512
g = torch.Generator(device=dev).manual_seed(9)
mp = torch.rand(N, 2, 512, W, device=dev, generator=g) * 2 - 1
mp[0, 1, 5, 0] = -0.0
bark = torch.rand(512, device=dev, generator=g) + 0.5
outs = []
for name in sys.argv[1:]:
    lib = ctypes.CDLL(name)
    lib.mg_codec_inv_ws_bytes.restype = ctypes.c_size_t; lib.mg_codec_inv_ws_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    f = lib.mg_codec_inv; f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    ws = torch.empty(lib.mg_codec_inv_ws_bytes(N, W), dtype=torch.uint8, device=dev)
    wav = torch.empty(256 * (N * W - 1), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: f(mp.data_ptr(), bark.data_ptr(), wav.data_ptr(), ws.data_ptr(), ws.numel(), N, W, s)
    rc = run()
    if rc != 0:
        lib.mg_last_error.restype = ctypes.c_char_p
        raise SystemExit(f"{name}: rc={rc}: {lib.mg_last_error().decode()}")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); e1.synchronize()
    print(name.split("/")[-1], f"{e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
    outs.append(wav.clone())
if len(outs) == 2:
    print("waveform bitwise equal:", torch.equal(outs[0], outs[1]), " max abs diff", float((outs[0] - outs[1]).abs().max()))
