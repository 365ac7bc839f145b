"""Per-kernel timings (HIP events, 20 launches each) of the layer shapes that dominate the level-5 batch-64 step.
Usage: python tools/bench_kernels.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)

def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters

rows = []
def conv_case(name, ci, co, h, w, ups=False, pn=False, lrelu=True, mask=False, dgrad=False):
    hin, win = (h // 2, w // 2) if ups else (h, w)
    x = R(N, ci, hin, win)
    wt = R(co, ci, 3, 3) * 0.05 if not dgrad else R(ci, co, 3, 3) * 0.05
    wp = ops.pack_conv3x3(wt, dgrad=dgrad)
    b = None if (mask or dgrad) else R(co)
    aux = R(N, co, h, w) if mask else None
    fn = lambda: ops.conv3x3(x, wp, b, co, ups=ups, lrelu=lrelu and not mask and not dgrad, mask_aux=aux, pixnorm=pn)
    ms = timeit(fn); fl = 2.0 * 9 * ci * co * h * w * N
    rows.append((name, ms, fl / ms / 1e9))

def wgrad_case(name, ci, co, h, w, ups=False):
    hin, win = (h // 2, w // 2) if ups else (h, w)
    x = R(N, ci, hin, win); gy = R(N, co, h, w)
    gw = torch.empty(co, ci, 3, 3, device=dev); gb = torch.empty(co, device=dev)
    fn = lambda: ops.conv3x3_wgrad(x, gy, gw, gb, ups=ups)
    ms = timeit(fn); fl = 2.0 * 9 * ci * co * h * w * N
    rows.append((name, ms, fl / ms / 1e9))

def bw_case(name, fn, nbytes):
    ms = timeit(fn); rows.append((name, ms, -nbytes / ms / 1e6))  # GB/s reported as negative marker

conv_case("G5.4 fwd ups+PN 64->48@128", 64, 48, 128, 128, ups=True, pn=True)
conv_case("G5.0 fwd PN 64->64@64", 64, 64, 64, 64, pn=True)
conv_case("G4.4 fwd ups+PN 80->64@64", 80, 64, 64, 64, ups=True, pn=True)
conv_case("D2.0 fwd lrelu 48->64@128", 48, 64, 128, 128)
conv_case("D2.0 tangent mask 48->64@128", 48, 64, 128, 128, mask=True)
conv_case("D2.3 fwd lrelu 64->64@64", 64, 64, 64, 64)
conv_case("D3.0 fwd lrelu 64->80@64", 64, 80, 64, 64)
conv_case("D2.0 dgrad+mask 64->48@128", 64, 48, 128, 128, mask=True, dgrad=True)
conv_case("G5.4 dgrad 48->64@128", 48, 64, 128, 128, dgrad=True)
conv_case("G4.0 fwd PN 80->80@32", 80, 80, 32, 32, pn=True)
conv_case("G3.0 fwd PN 96->96@16", 96, 96, 16, 16, pn=True)
conv_case("D5.3 fwd 112->112@8", 112, 112, 8, 8)
conv_case("D6.3 fwd 128->128@4", 128, 128, 4, 4)
conv_case("D7.3 fwd 144->144@2", 144, 144, 2, 2)
conv_case("D8.3 fwd 160->160@1", 160, 160, 1, 1)
wgrad_case("G5.4 wgrad ups 64->48@128", 64, 48, 128, 128, ups=True)
wgrad_case("D2.0 wgrad 48->64@128", 48, 64, 128, 128)
wgrad_case("D2.3 wgrad 64->64@64", 64, 64, 64, 64)
wgrad_case("D3.0 wgrad 64->80@64", 64, 80, 64, 64)
wgrad_case("G3.0 wgrad 96->96@16", 96, 96, 16, 16)
wgrad_case("D6.3 wgrad 128->128@4", 128, 128, 4, 4)
wgrad_case("D8.3 wgrad 160->160@1", 160, 160, 1, 1)
y = R(N, 48, 128, 128); rn = torch.rand(N, 1, 128, 128, device=dev) + 0.5; gp = R(N, 48, 128, 128)
bw_case("pixelnorm_lrelu_bwd 48@128", lambda: ops.pixelnorm_lrelu_bwd(gp, y, rn), 3 * y.numel() * 4)
a1 = R(N, 64, 128, 128); gq = R(N, 64, 64, 64)
bw_case("avgpool2_bwd+mask 64@128", lambda: ops.avgpool2_bwd(gq, a1), (2 * a1.numel() + gq.numel()) * 4)
bw_case("avgpool2_fwd 64@128", lambda: ops.avgpool2_fwd(a1), (a1.numel() + gq.numel()) * 4)
x2 = R(N, 2, 128, 128); ws = R(48, 2, 1, 1); bs = R(48)
bw_case("stem 2->48@128 lrelu", lambda: ops.conv1x1(x2, ws, bs, 48, lrelu=True), (x2.numel() + y.numel()) * 4)
wh = R(2, 48, 1, 1); bh = R(2)
bw_case("head 48->2@128 tanh", lambda: ops.conv1x1(y, wh, bh, 2, tanh=True), (x2.numel() + y.numel()) * 4)
gw1 = torch.empty(48, 2, 1, 1, device=dev); gb1 = torch.empty(48, device=dev)
bw_case("stem wgrad 2->48@128", lambda: ops.conv1x1_wgrad(x2, y, gw1, gb1), (x2.numel() + y.numel()) * 4)
bw_case("axpby 64@64", lambda: ops.axpby(0.5, gq, 0.5, gq), 3 * gq.numel() * 4)
print(f"{'kernel':42s} {'ms':>8s} {'TFLOP/s | GB/s':>16s}")
for name, ms, v in rows:
    print(f"{name:42s} {ms:8.3f} {('%.1f TF' % v) if v >= 0 else ('%.0f GB/s' % -v):>16s}")

# sub-pixel form of the generator's upsample-convs vs the direct form (same layers)
rows.clear()
def up_case(name, ci, co, hin):
    x = R(N, ci, hin, hin); wt = R(co, ci, 3, 3) * 0.05; b = R(co)
    wpu = ops.pack_upconv3x3(wt); wpd = ops.pack_conv3x3(wt, dgrad=False)
    fl = 2.0 * 9 * ci * co * (2 * hin) ** 2 * N
    ms = timeit(lambda: ops.upconv3x3(x, wpu, b, co, lrelu=True, pixnorm=True, want_y=False))
    rows.append((name + " sub-pixel", ms, fl / ms / 1e9))
    ms = timeit(lambda: ops.conv3x3(x, wpd, b, co, ups=True, lrelu=True, pixnorm=True, want_y=False))
    rows.append((name + " direct", ms, fl / ms / 1e9))
up_case("G5.4 64->48 @64->128", 64, 48, 64)
up_case("G4.4 80->64 @32->64", 80, 64, 32)
up_case("G3.4 96->80 @16->32", 96, 80, 16)
up_case("G2.4 112->96 @8->16", 112, 96, 8)
print("(TF/s below are ALGORITHMIC: 18*Cin*Cout per output pixel)")
for name, ms, v in rows:
    print(f"{name:42s} {ms:8.3f} {('%.1f TF' % v):>16s}")
