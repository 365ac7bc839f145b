"""Step time of (level, batch) under each kernel-selection threshold scaled by 1/4 .. 4 (the thresholds were tuned at levels 4-5 and
batches 32-64).  python tools/sweep_thresholds.py 6:6 7:6 5:64"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARS = {"MG_WINO_MIN_PIXELS": 8192, "MG_WINO_WGRAD_MIN_PIXELS": 64, "MG_PN_FUSE_MIN_PIXELS": 16384,
        "MG_UPCONV_DGRAD_MIN_PIXELS": 16384, "MG_UPCONV_MIN_LOWRES_PIXELS": 32768}
cases = [tuple(map(int, c.split(":"))) for c in (sys.argv[1:] or ["6:6", "7:6", "5:64"])]

def run(level, batch, env):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--level", str(level), "--batch", str(batch), "--steps", "30",
                          "--warmup", "10", "--no-extra", "--no-cpu-baseline"], capture_output=True, text=True, env=e)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return json.loads(line[-1])["ms_per_step"] if line else float("nan")

for (level, batch) in cases:
    base = run(level, batch, {})
    print(f"== level {level} batch {batch}: default {base:.3f} ms", flush=True)
    for var, dflt in VARS.items():
        res = []
        for f in (0.25, 0.5, 2, 4):
            ms = run(level, batch, {var: str(int(dflt * f))})
            res.append(f"x{f}: {ms:.3f}")
        print(f"   {var:30s} " + "  ".join(res), flush=True)
