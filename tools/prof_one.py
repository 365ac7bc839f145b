"""Run ONE kernel shape a few times (for rocprofv3 --pmc / --kernel-trace).  python3 tools/prof_one.py <case> [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops
case = sys.argv[1] if len(sys.argv) > 1 else "g54"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = 64
dev = torch.device("cuda", 0)
R = lambda *s: torch.randn(*s, device=dev)
if case == "g54":      # fused ups conv 64->48 @128 + lrelu + pixnorm
    x = R(N, 64, 64, 64); wp = ops.pack_conv3x3(R(48, 64, 3, 3) * 0.05, False); b = R(48)
    fn = lambda: ops.conv3x3(x, wp, b, 48, ups=True, lrelu=True, pixnorm=True, want_y=False)
elif case == "up54":   # the generator's last up-sampling conv in sub-pixel form: 64 -> 48, 64x64 -> 128x128, LeakyReLU + PixelNorm
    x = R(N, 64, 64, 64); wp = ops.pack_upconv3x3(R(48, 64, 3, 3) * 0.05); b = R(48)
    fn = lambda: ops.upconv3x3(x, wp, b, 48, lrelu=True, pixnorm=True, want_y=False)
elif case == "updg54":  # its data gradient: (N, 48, 128, 128) -> (N, 64, 64, 64)
    gy = R(N, 48, 128, 128); wp = ops.pack_upconv3x3_dgrad(R(48, 64, 3, 3) * 0.05)
    fn = lambda: ops.upconv3x3_dgrad(gy, wp, 64)
elif case == "wu54":   # the same layer in 9-component Winograd form (wino_ups.hip)
    x = R(N, 64, 64, 64); up = ops.pack_winoups3x3(R(48, 64, 3, 3) * 0.05, False); b = R(48)
    fn = lambda: ops.winoups3x3(x, up, b, 48, lrelu=True, pixnorm=True, want_y=False)
elif case == "wudg54":  # and its data gradient
    gy = R(N, 48, 128, 128); up = ops.pack_winoups3x3(R(48, 64, 3, 3) * 0.05, True)
    fn = lambda: ops.winoups3x3_dgrad(gy, up, 64)
elif case == "d20":    # conv 48->64 @128 + lrelu
    x = R(N, 48, 128, 128); wp = ops.pack_conv3x3(R(64, 48, 3, 3) * 0.05, False); b = R(64)
    fn = lambda: ops.conv3x3(x, wp, b, 64, lrelu=True)
elif case == "w20":    # wgrad 48->64 @128
    x = R(N, 48, 128, 128); gy = R(N, 64, 128, 128); gw = torch.empty(64, 48, 3, 3, device=dev); gb = torch.empty(64, device=dev)
    fn = lambda: ops.conv3x3_wgrad(x, gy, gw, gb)
elif case == "w20n":   # the weight gradient of the step's dominant layer as the critic update runs it: 48 x 64 @128 over 3N = 192 images
    x = R(3 * N, 48, 128, 128); gy = R(3 * N, 64, 128, 128); gw = torch.empty(64, 48, 3, 3, device=dev); gb = torch.empty(64, device=dev)
    fn = lambda: ops.conv3x3_wgrad(x, gy, gw, gb)
elif case == "w54":
    x = R(N, 64, 64, 64); gy = R(N, 48, 128, 128); gw = torch.empty(48, 64, 3, 3, device=dev); gb = torch.empty(48, device=dev)
    fn = lambda: ops.conv3x3_wgrad(x, gy, gw, gb, ups=True)
elif case == "wino3n":  # the step's dominant launch: Winograd conv 48->64 @128 + lrelu + fused pool + tile mask over 3N images
    x = R(3 * N, 48, 128, 128); up = ops.pack_wino3x3(R(64, 48, 3, 3) * 0.05, False); b = R(64)
    q = torch.empty(3 * N, 64, 64, 64, device=dev)
    fn = lambda: ops.conv3x3(x, None, b, 64, lrelu=True, pool_out=q, wino=up, mask_out=True)
elif case == "wino":   # Winograd conv 48->64 @128 + lrelu + fused pool
    x = R(N, 48, 128, 128); up = ops.pack_wino3x3(R(64, 48, 3, 3) * 0.05, False); b = R(64)
    fn = lambda: ops.conv3x3(x, None, b, 64, lrelu=True, pool=True, wino=up)
elif case == "stft":
    wav = torch.rand(44100 * 600, device=dev) - 0.5
    fn = lambda: ops.stft_1024(wav)
elif case == "ww16":   # level-7 critic, first conv: weight gradient 16 x 32 channels @512x512 over 3N = 18 images
    x = R(18, 16, 512, 512); gy = R(18, 32, 512, 512); gw = torch.empty(32, 16, 3, 3, device=dev); gb = torch.empty(32, device=dev)
    fn = lambda: ops.conv3x3_wgrad(x, gy, gw, gb)
elif case == "dg16":   # level-7 critic, first conv: data gradient 32 -> 16 channels @512x512 over 18 images, LeakyReLU mask
    x = R(18, 32, 512, 512); up = ops.pack_wino3x3(R(16, 32, 3, 3) * 0.05, False); aux = R(18, 16, 512, 512)
    fn = lambda: ops.conv3x3(x, None, None, 16, mask_aux=aux, wino=up)
elif case == "fw16":   # level-7 critic, first conv forward: 16 -> 32 @512x512 + lrelu + pool over 18 images
    x = R(18, 16, 512, 512); up = ops.pack_wino3x3(R(32, 16, 3, 3) * 0.05, False); b = R(32)
    fn = lambda: ops.conv3x3(x, None, b, 32, lrelu=True, pool=True, wino=up)
elif case == "fwm16":  # the same layer as the critic's forward pass runs it: + tile mask out, the full-resolution y never written
    x = R(18, 16, 512, 512); up = ops.pack_wino3x3(R(32, 16, 3, 3) * 0.05, False); b = R(32)
    fn = lambda: ops.conv3x3(x, None, b, 32, lrelu=True, pool=True, wino=up, mask_out=True)
elif case == "codec":
    from musicgan_amd import audio
    c = ops.stft_1024(torch.rand(44100 * 600, device=dev) - 0.5)
    fn = lambda: audio.stft_to_phase_magn(c)
else:
    raise SystemExit("unknown case")
for _ in range(iters):
    fn()
torch.cuda.synchronize()
