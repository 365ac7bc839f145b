"""GPU: the data-parallel code path of the training step (side HIP stream, flat gradient bucket, RCCL all-reduce, fused Adam queued
behind it, per-rank noise) on ONE GPU -- a 1-rank "nccl" process group with MG_FORCE_DP=1 runs exactly the code N ranks run.
The N > 1 arithmetic (sum of per-rank gradients == concatenated-batch gradient) is covered on CPU by tests/test_dist_cpu.py."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def one_rank_nccl(monkeypatch):
    import torch.distributed as dist
    monkeypatch.setenv("MG_FORCE_DP", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(29650 + os.getpid() % 200))
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    yield
    dist.destroy_process_group()


def _run(steps, level=3, batch=4):
    import bench
    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train_step import ProGANStepper
    gen, disc = bench.build_nets(level, 32, DEV)
    og = FusedAdam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9))
    od = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))
    st = ProGANStepper(gen, disc, og, od, 32)
    side = bench.LEVEL_SIDE[level]
    rng = torch.Generator(device=DEV).manual_seed(77)
    x_real = torch.rand(batch, 2, side, side, device=DEV, generator=rng) * 2 - 1
    losses = []
    for i in range(steps):
        z = torch.randn(batch, 32, 2, 2, device=DEV, generator=rng)
        eps = torch.rand(batch, 1, 1, 1, device=DEV, generator=rng)
        z2 = torch.randn(batch, 32, 2, 2, device=DEV, generator=rng)
        d = st.d_step(x_real, 0.5, z=z, eps=eps)
        losses.append(d["disc_loss"])
        if i % 2 == 0:  # both orders of the overlap: D step after a G step, and D step after a D step
            losses.append(st.g_step(batch, 0.5, DEV, z=z2)["gen_loss"])
    st.finish()
    torch.cuda.synchronize()
    weights = {("G." if net is gen else "D.") + k: p.detach().clone() for net in (gen, disc) for k, p in net.named_parameters()}
    return st, weights, [float(v) for v in losses]


def test_data_parallel_path_is_bit_identical_to_single_process(one_rank_nccl, monkeypatch):
    """Six critic + three generator updates through the final stepper (fused critic step; the third call of each kind captures
    and replays its HIP graph -- under DP the graph ends with the gradients in the flat bucket and the exchange + Adam follow it on
    the side stream) with the DP machinery live, against the same run without it: every weight bit-identical, every loss equal."""
    st_dp, w_dp, l_dp = _run(6)
    assert st_dp.use_graphs and len(st_dp._graphs) == 2 and all("graph" in e for e in st_dp._graphs.values())
    assert st_dp.dp and st_dp.bucket_d.stream() is not None and st_dp.bucket_g.stream() is not None
    assert st_dp.optim_disc.grad_scale == 1.0
    monkeypatch.setenv("MG_FORCE_DP", "0")
    st_sp, w_sp, l_sp = _run(6)
    assert not st_sp.dp
    assert l_dp == l_sp
    assert w_dp.keys() == w_sp.keys()
    for k in w_dp:
        assert torch.equal(w_dp[k], w_sp[k]), k
    # gradients of the DP run are views into ONE flat buffer per network (what was all-reduced)
    flat = st_dp.bucket_d._flat  # the run's last update was a critic step (a graph replay)
    live = [p for p in st_dp.disc.parameters() if p.grad is not None]
    assert flat is not None and sum(p.numel() for p in live) == flat.numel()
    assert all(flat.data_ptr() <= p.grad.data_ptr() < flat.data_ptr() + 4 * flat.numel() for p in live)
    # ... for BOTH networks, and neither exchange flattened by copy (the generator's parameter order differs from its engine layout:
    # a consecutive-order test used to fail there and every generator update paid a torch.cat on the side stream)
    assert st_dp.bucket_d.copied is False and st_dp.bucket_g.copied is False
    assert st_dp.bucket_g._flat is st_dp.bucket_g._own and st_dp.bucket_d._flat is st_dp.bucket_d._own


def test_bench_runs_under_torchrun_as_the_driver_launches_it(tmp_path):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` (the driver's N > 1 command) with
    N = 1 and MG_FORCE_DP=1: a 1-rank RCCL group, the data-parallel stepper, one JSON line with the contract's keys."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MG_FORCE_DP="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29800 + os.getpid() % 100), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "2", "--level", "3", "--batch", "8", "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["parallelism"] == "dp1" and "workload" in d["config"]


def test_two_ranks_sharing_the_gpu_equal_one_process_on_the_concatenated_batch(tmp_path):
    """N = 2 for real: two processes (gloo -- both ranks on this box's one GPU, which RCCL does not allow; everything above the
    backend is the product's DP path: flat bucket, side stream, exchange, fused Adam with 1/world, HIP-graph replay of the
    gradient computation from the third update on) each take half of a fixed batch; the mean gradient of every critic update
    must equal the single-process gradient on the whole batch within fp32 summation noise."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = os.path.join(root, "tests", "dp_two_rank_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MG_FORCE_DP="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    r = subprocess.run([sys.executable, worker, one], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + os.getpid() % 90), worker, two]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    g1, g2 = torch.load(one), torch.load(two)
    assert len(g1) == len(g2) == 4
    for step, (a, b) in enumerate(zip(g1, g2)):
        assert a.keys() == b.keys() and len(a) > 10
        for k in a:
            scale = float(a[k].abs().max()) + 1e-12
            err = float((a[k] - b[k]).abs().max())
            assert err <= 2e-4 * scale + 1e-7, f"update {step} {k}: {err:.2e} vs max {scale:.2e}"


def test_bench_with_two_ranks_sharing_the_gpu(tmp_path):
    """The driver's N = 2 command (`torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`) rehearsed on this box's one
    GPU (MG_BENCH_SHARE_GPU=1: both ranks on cuda:0 over gloo): barriers, the max-over-ranks timing, the data-parallel stepper
    with graph replay under two real ranks, rank 0 printing ONE line whose value is the whole-job rate."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MG_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", MG_FORCE_DP="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 90), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "4", "--level", "3", "--batch", "8", "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert d["config"]["global_batch"] == 16
    assert abs(d["value"] - 2 * 8 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]  # whole-job images/s over the slowest rank's time
