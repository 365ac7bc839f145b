"""Per-kernel table of a rocprofv3 --kernel-trace CSV (steady-state tail of the run), grouped by kernel name and workgroup count.
    python tools/trace_table.py TRACE.csv [top] [BENCH.json of the same (profiled) run: its ms_per_step is printed beside the busy
    time] [BENCH.json of an un-profiled run of the same configuration]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the steps end with the last fused-Adam launch: what bench.py runs behind them (the dominant-kernel probe: 130 launches) is not a step
last = max(i for i, r in enumerate(rows) if 'adam_dev_k' in r['Kernel_Name'])
rows = rows[:last + 1]
sub = rows[int(len(rows) * 0.55):]
steps = sum('adam_dev_k' in r['Kernel_Name'] for r in sub) / 2.0
span = int(sub[-1]['End_Timestamp']) - int(sub[0]['Start_Timestamp'])
c = collections.defaultdict(lambda: [0, 0])
for r in sub:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    nm = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    nm = re.sub(r'\(.*', '', nm)[:48]
    g = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) * (int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y'])))
    c[(nm, g)][0] += 1
    c[(nm, g)][1] += d
tot = sum(v[1] for v in c.values())
line = f"D+G steps {steps:.1f}, kernels/step {len(sub) / steps:.0f}, busy {tot / steps / 1e6:.3f} ms/step, span {span / steps / 1e6:.3f} ms/step"
if len(sys.argv) > 3:
    import json
    d = json.loads([ln for ln in open(sys.argv[3]) if ln.startswith("{")][-1])
    line += f"; ms_per_step of the same (profiled) run {d['ms_per_step']:.3f} (busy / that = {tot / steps / 1e6 / d['ms_per_step']:.3f})"
if len(sys.argv) > 4:  # the same configuration on the same box right before, without the profiler
    u = json.loads([ln for ln in open(sys.argv[4]) if ln.startswith("{")][-1])
    line += f"; un-profiled run just before on the same box {u['ms_per_step']:.3f} ms (busy / that = {tot / steps / 1e6 / u['ms_per_step']:.3f})"
print(line)
for (k, g), (cnt, t) in sorted(c.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"  {k:50s} WGs {g:6d} {cnt / steps:5.1f}/step {t / cnt / 1e3:7.1f} us  {t / steps / 1e3:7.1f} us/step {100 * t / tot:5.1f}%")
