import ctypes, sys, torch
dev=torch.device("cuda",0)
L=44100*600; T=1+L//256
wav=torch.rand(L,device=dev)-0.5
out=torch.empty(512,T,2,device=dev)
for name in sys.argv[1:]:
    lib=ctypes.CDLL(name)
    f=lib.mg_stft_1024; f.restype=ctypes.c_int; f.argtypes=[ctypes.c_void_p]*3+[ctypes.c_int64, ctypes.c_void_p]
    s=torch.cuda.current_stream().cuda_stream
    for _ in range(3): f(wav.data_ptr(), out.data_ptr(), None, L, s)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f(wav.data_ptr(), out.data_ptr(), None, L, s)
    e1.record(); e1.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(name.split('/')[-1], f"{ms:.4f} ms  {T/ms/1e3:.1f} M frames/s", flush=True)
