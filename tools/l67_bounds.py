"""Per-launch bound table of one D+G step (levels 6 / 7 at the reference's batch 6, train.py:43): every `musicgan_amd.ops` call timed on
its own with HIP events (device sync after each, graphs off, weight gradients one launch per layer: MG_WGRAD_GROUP=0) beside
    bound = max(algorithmic bytes / 8 TB/s, executed FLOP / 157.3 TFLOP/s)
algorithmic bytes = every tensor the call reads + every tensor it writes, once each; executed FLOP = 18 Cin Cout per output pixel for
a 3x3 convolution pass, / 2.25 where it runs in Winograd or sub-pixel form, 2 Cin Cout for 1x1.  Calls of >= 60 us are listed.
    python tools/l67_bounds.py [level] [batch] [reps] > profiles/r05_l67_bounds_l<level>.txt"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MG_GRAPHS"] = "0"
os.environ["MG_WGRAD_GROUP"] = "0"
import torch  # noqa: E402

import bench  # noqa: E402
from musicgan_amd import ops  # noqa: E402
from musicgan_amd.optim import FusedAdam  # noqa: E402
from musicgan_amd.train_step import ProGANStepper  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 7
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
floor_us = float(sys.argv[4]) if len(sys.argv) > 4 else 60.0
dev = torch.device("cuda", 0)
records = collections.OrderedDict()
phase = ["warm"]
_orig = {}


def tensors(v, acc):
    if isinstance(v, torch.Tensor):
        acc.append(v)
    elif isinstance(v, (tuple, list)):
        for x in v:
            tensors(x, acc)


def flops(name, a, k, res):
    """Executed FLOPs of the call (see the module docstring)."""
    W = 1.0 / 2.25
    if name in ("conv3x3", "conv3x3_fade", "conv3x3_small", "conv3x3_small_pn"):
        x = a[0]
        n, ci, h, w = x.shape
        co = a[3]
        if k.get("ups"):
            h, w = 2 * h, 2 * w
        wino = name == "conv3x3_fade" or k.get("wino") is not None
        return 18.0 * n * ci * co * h * w * (W if wino else 1.0)
    if name == "upconv3x3":
        n, ci, h, w = a[0].shape
        return 18.0 * n * ci * a[3] * 4 * h * w * W
    if name in ("winoups3x3", "winoups3x3_head"):  # 9 of 36 multiply-adds per output pixel and channel pair
        n, ci, h, w = a[0].shape
        return 18.0 * n * ci * a[3] * 4 * h * w * 0.25
    if name == "winoups3x3_dgrad":
        n, co, h2, w2 = a[0].shape
        return 18.0 * n * a[2] * co * h2 * w2 * 0.25
    if name == "upconv3x3_dgrad":
        n, co, h2, w2 = a[0].shape
        return 18.0 * n * a[2] * co * h2 * w2 * W
    if name == "conv3x3_wgrad":
        x, gy = a[0], a[1]
        n, co, h, w = gy.shape
        ci = x.shape[1]
        return 18.0 * n * ci * co * h * w * (W if _orig["wino_wgrad_supported"](n, ci, co, h, w, ups=k.get("ups", False)) else 1.0)
    if name in ("conv1x1", "conv1x1_wgrad"):
        x = a[0]
        n, ci, h, w = x.shape
        co = a[3] if name == "conv1x1" else a[1].shape[1]
        return 2.0 * n * ci * co * h * w
    return 0.0


def wrap(name, fn):
    def w(*a, **k):
        ins = []
        tensors(a, ins)
        tensors([v for kk, v in k.items() if kk not in ("out", "pool_out")], ins)
        outs_kw = []
        tensors([k.get("out"), k.get("pool_out")], outs_kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()  # launches that are not wrapped (packs, reduces, Adam) must not land in this call's time
        e0.record()
        r = fn(*a, **k)
        e1.record()
        e1.synchronize()
        outs = []
        tensors(r, outs)
        outs += outs_kw
        if name in ("conv3x3_wgrad", "conv1x1_wgrad"):  # gw / gb are written, not read
            outs += [t for t in ins[2:4]]
            ins = ins[:2] + ins[4:]
        seen, nbytes = set(), 0
        for side, group in (("r", ins), ("w", outs)):  # an in-place call reads AND writes that storage: counted on both sides
            for t in group:
                key = (side, t.data_ptr(), t.numel())
                if key not in seen:
                    seen.add(key)
                    nbytes += t.numel() * t.element_size()
        shp = " ".join("x".join(map(str, t.shape)) for t in ins[:2] if t.dim() == 4)
        flags = ",".join(sorted(kk for kk, vv in k.items() if (vv is True) or (isinstance(vv, torch.Tensor) and kk in
                                                                               ("mask_aux", "unpool_mask", "wino", "mask_in", "tanh_y", "unpool_aux"))
                                or (kk == "mask_out" and vv)))
        co = next((str(x) for x in a if isinstance(x, int)), "")
        key = (phase[0], name, shp, co + " " + flags)
        rec = records.setdefault(key, [0, 0.0, nbytes, flops(name, a, k, r)])
        rec[0] += 1
        rec[1] += e0.elapsed_time(e1)
        return r
    return w


_orig["wino_wgrad_supported"] = ops.wino_wgrad_supported
for n in ("conv3x3", "conv3x3_fade", "conv3x3_small", "conv3x3_small_pn", "upconv3x3", "upconv3x3_dgrad", "conv3x3_wgrad", "conv1x1",
          "conv1x1_wgrad", "pixelnorm_fwd", "pixelnorm_lrelu_bwd", "upsample2x_fwd", "upsample2x_bwd", "avgpool2_fwd", "avgpool2_bwd",
          "blend_lrelu_bwd", "lrelu_bwd", "axpby", "blend_up", "gp_interp", "sumsq_per_sample", "scale_per_sample", "winoups3x3",
          "winoups3x3_head", "winoups3x3_dgrad", "gen_head_bwd", "head_pair", "head_pair_from_mp", "stem_pair", "stem_pair_gx",
          "blend_up_bwd", "gp_apply", "linear1_fwd", "linear1_bwd"):
    setattr(ops, n, wrap(n, getattr(ops, n)))

# the slab reductions of a sweep's weight gradients are ONE launch for all its layers (WgradDefer.flush): a row of its own, `group`
_flush = ops.WgradDefer.flush


def flush_timed(self):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    _flush(self)
    e1.record()
    e1.synchronize()
    rec = records.setdefault((phase[0], "wgrad_reduce", "group: all layers of the sweep", ""), [0, 0.0, 0, 0.0])
    rec[0] += 1
    rec[1] += e0.elapsed_time(e1)


ops.WgradDefer.flush = flush_timed
gen, disc = bench.build_nets(level, 32, dev)
og = FusedAdam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9))
od = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))
st = ProGANStepper(gen, disc, og, od, 32)
side = bench.LEVEL_SIDE[level]
rng = torch.Generator(device=dev).manual_seed(1)
x_real = torch.rand(batch, 2, side, side, device=dev, generator=rng) * 2 - 1
for i in range(reps + 1):
    if i == 1:
        records.clear()
    phase[0] = "D"
    st.d_step(x_real, 0.5)
    phase[0] = "G"
    st.g_step(batch, 0.5, dev)
tot = sum(ms for (_, ms, _, _) in records.values()) / reps
print(f"# level {level} batch {batch}: {len(records)} distinct calls, per-call sum {tot:.3f} ms per D+G step (each call synchronised; the step "
      f"itself replays as a graph in less)")
print(f"# calls of >= {floor_us:.0f} us; bound = max(bytes / 8 TB/s, executed FLOP / 157.3 TF/s); D = critic update, G = generator update")
print("# step op                  tensors (first two 4-D inputs)            cout flags                        calls  us/call  MB      GFLOP   bound us  x bound  (limit)")
rows = sorted(records.items(), key=lambda kv: -kv[1][1] / kv[1][0])
if os.environ.get("MG_BOUNDS_ORDER") == "program":  # the calls in the order the step makes them (first occurrence)
    rows = list(records.items())
over, listed = 0.0, 0.0
for (ph, name, shp, flags), (cnt, ms, nbytes, fl) in rows:
    us = 1e3 * ms / cnt
    if us < floor_us:
        continue
    tb, tf = nbytes / 8e12 * 1e6, fl / 157.3e12 * 1e6
    bound = max(tb, tf)
    if bound <= 0.0:  # (a group row: no bound of its own)
        listed += us * cnt / reps
        print(f"{ph} {name:20s} {shp:42s} {flags:34s} {cnt / reps:5.1f} {us:8.1f}       -       -        -       -")
        continue
    listed += us * cnt / reps
    over += max(0.0, us - 1.8 * bound) * cnt / reps
    print(f"{ph} {name:20s} {shp:42s} {flags:34s} {cnt / reps:5.1f} {us:8.1f} {nbytes / 1e6:7.1f} {fl / 1e9:7.2f} {bound:8.1f} {us / bound:7.2f}  ({'hbm' if tb >= tf else 'mfma'})")
print(f"# listed calls: {listed / 1e3:.3f} ms per step; time above 1.8 x bound: {over / 1e3:.3f} ms per step")
