"""Per-call attribution of one D step + one G step: every `musicgan_amd.ops` call timed on its own with HIP events (a device sync
after each call, so launch gaps are excluded) and keyed by op name + tensor shapes + fused flags.
    python tools/op_profile.py [level] [batch] [reps]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MG_GRAPHS"] = "0"  # every call is followed by a device sync here: not capturable
import torch  # noqa: E402

import bench  # noqa: E402
from musicgan_amd import ops  # noqa: E402
from musicgan_amd.optim import FusedAdam  # noqa: E402
from musicgan_amd.train_step import ProGANStepper  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
records = collections.OrderedDict()
phase = ["warm"]


def wrap(name, fn):
    def w(*a, **k):
        def desc(v):
            if isinstance(v, torch.Tensor):
                return "x".join(map(str, v.shape))
            return None
        shapes = [d for d in map(desc, a) if d][:2]
        flags = [kk for kk, vv in k.items() if (vv is not None and vv is not False and kk not in ("out", "pool_out", "want_y"))
                 and not isinstance(vv, (torch.Tensor, float, int)) or (isinstance(vv, torch.Tensor) and kk in ("mask_aux", "tanh_bwd_in", "wino", "tanh_y"))
                 or (vv is True)]
        extra = [str(x) for x in a if isinstance(x, int)][:1]
        key = (phase[0], name, " ".join(shapes), ",".join(sorted(set(flags)) + extra))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        e1.synchronize()
        rec = records.setdefault(key, [0, 0.0])
        rec[0] += 1
        rec[1] += e0.elapsed_time(e1)
        return r
    return w


for n in ("conv3x3", "upconv3x3", "upconv3x3_dgrad", "conv3x3_wgrad", "conv1x1", "conv1x1_wgrad", "pixelnorm_lrelu_bwd",
          "upsample2x_bwd", "avgpool2_fwd", "avgpool2_bwd", "blend_lrelu_bwd", "lrelu_bwd", "axpby", "blend_up", "linear1_fwd",
          "linear1_bwd", "gp_interp", "sumsq_per_sample", "scale_per_sample", "gp_finish", "pack_conv3x3", "pack_wino3x3",
          "pack_upconv3x3", "pack_upconv3x3_dgrad"):
    setattr(ops, n, wrap(n, getattr(ops, n)))

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
tot = {"D": 0.0, "G": 0.0}
for (ph, name, shapes, flags), (cnt, ms) in records.items():
    tot[ph] += ms / reps
print(f"level {level} batch {batch}: per-call sum D step {tot['D']:.3f} ms, G step {tot['G']:.3f} ms (each call synchronised)")
rows = sorted(records.items(), key=lambda kv: -kv[1][1])
for (ph, name, shapes, flags), (cnt, ms) in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 70]:
    print(f"{ph} {name:22s} {shapes:34s} {flags:32s} calls/step {cnt / reps:5.1f}  ms/step {ms / reps:7.3f}  us/call {1e3 * ms / cnt:7.1f}")
