"""Per-kernel table of a rocprofv3 --kernel-trace CSV (steady-state tail of the run), grouped by kernel name and workgroup count.
    python tools/trace_table.py TRACE.csv [top]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r['Start_Timestamp']))
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
print(f"D+G steps {steps:.1f}, kernels/step {len(sub) / steps:.0f}, busy {tot / steps / 1e6:.3f} ms/step, span {span / steps / 1e6:.3f} ms/step")
for (k, g), (cnt, t) in sorted(c.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"  {k:50s} WGs {g:6d} {cnt / steps:5.1f}/step {t / cnt / 1e3:7.1f} us  {t / steps / 1e3:7.1f} us/step {100 * t / tot:5.1f}%")
