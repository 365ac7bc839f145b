"""Row-staged weight-gradient kernel, one shape, us per call under the measurement switches given in the environment.
python tools/ab_wgrad_rows.py [N Cin Cout H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops
dev = torch.device("cuda", 0)
n, ci, co, h = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (192, 48, 64, 128)
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(n, ci, h, h, device=dev, generator=g); gy = torch.randn(n, co, h, h, device=dev, generator=g)
gw, gb = torch.empty(co, ci, 3, 3, device=dev), torch.empty(co, device=dev)
fn = lambda: ops.conv3x3_wgrad(x, gy, gw, gb)
for _ in range(5): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); e1.synchronize()
print(f"{n} {ci}x{co} @{h}  ROWS={os.environ.get('MG_WGRAD_ROWS','1')} ABLATE={os.environ.get('MG_WGRAD_ROWS_ABLATE','0')}  {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us  checksum {float(gw.double().abs().sum()):.6e}", flush=True)
