"""One D+G step of a rocprofv3 --kernel-trace CSV as a SEQUENCE (steady-state: the last full step of the run): per launch the
kernel name, workgroups, duration and the idle gap in front of it.
    python tools/trace_seq.py TRACE.csv [skip_steps_from_end]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step = [pack .. adam] critic update + [pack .. adam] generator update; cut at adam_dev_k launches
adam = [i for i, r in enumerate(rows) if 'adam_dev_k' in r['Kernel_Name']]
assert len(adam) >= 2 * back + 3, "trace too short"
lo, hi = adam[-2 * back - 3] + 1, adam[-2 * back - 1] + 1
sub = rows[lo:hi]
t_prev = int(rows[lo - 1]['End_Timestamp'])
busy = gaps = 0
print(f"{len(sub)} launches")
for r in sub:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    nm = re.sub(r'\(.*', '', nm).replace('void ', '')[:44]
    g = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) * (int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y']))) * \
        (int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_Z'])))
    print(f"{nm:46s} WGs {g:6d} x{int(r['Workgroup_Size_X']):4d}  {(e - s) / 1e3:7.1f} us  gap {(s - t_prev) / 1e3:6.1f}")
    busy += e - s
    gaps += max(0, s - t_prev)
    t_prev = max(t_prev, e)
print(f"busy {busy / 1e6:.3f} ms, gaps {gaps / 1e6:.3f} ms, span {(int(sub[-1]['End_Timestamp']) - int(sub[0]['Start_Timestamp'])) / 1e6:.3f} ms")
