"""Diagnostics for tests/test_dp_gpu.py's two-rank comparison: run-to-run determinism of the single-process worker, MG_SMALLNET 0 vs 1,
and the worst error / tolerance ratios of one rank vs two (diagnostic)."""
import os, subprocess, sys, tempfile
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
worker = os.path.join(root, "tests", "dp_two_rank_worker.py")
d = tempfile.mkdtemp()

def run(sn, two, tag, seed="5"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MG_FORCE_DP="0", MG_SMALLNET=sn, MG_TEST_SEED=seed)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = os.path.join(d, tag + ".pt")
    cmd = [sys.executable, worker, out] if not two else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
          "--master-addr", "127.0.0.1", "--master-port", "29931", worker, out]
    subprocess.run(cmd, check=True, env=env, cwd=root, capture_output=True)
    return torch.load(out)

def cmp(a_, b_, what):
    rows = []
    for step, (a, b) in enumerate(zip(a_, b_)):
        for k in a:
            scale = float(a[k].abs().max()) + 1e-12
            err = float((a[k] - b[k]).abs().max())
            rows.append((err / (2e-4 * scale + 1e-7), err, scale, step, k))
    rows.sort(reverse=True)
    print(what)
    for r in rows[:5]:
        print("   ratio %.3f err %.2e scale %.2e update %d %s" % r)

for seed in ("5", "6", "7", "8", "9", "10"):
    for sn in ("0", "1"):
        cmp(run(sn, False, "a", seed), run(sn, True, "b", seed), f"seed {seed} MG_SMALLNET={sn}: one rank vs two")
