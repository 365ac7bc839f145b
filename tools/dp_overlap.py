"""Where the data-parallel exchange runs: from a rocprofv3 --kernel-trace CSV of a 1-rank RCCL run (MG_FORCE_DP=1), for every all-reduce
kernel and every adam_dev_k launch of the steady-state tail: its queue / stream, its start / end relative to the update, and the
kernels of the MAIN queue that run while it does (the next update's generator forward).
    python tools/dp_overlap.py TRACE.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
qcol = next((c for c in ("Stream_Id", "Queue_Id") if c in rows[0]), None)
assert qcol, f"no queue / stream column in {list(rows[0].keys())}"
name = lambda r: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])).replace("void ", "")[:40]
tail = rows[len(rows) // 2:]
counts = {}
for r in tail:
    counts[r[qcol]] = counts.get(r[qcol], 0) + 1
main_q = max(counts, key=counts.get)
print(f"column {qcol}: launches per id in the second half of the run {counts}; main = {main_q}")
side = [r for r in tail if r[qcol] != main_q]
shown = 0
for r in side:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    over = [m for m in tail if m[qcol] == main_q and int(m["Start_Timestamp"]) < e and int(m["End_Timestamp"]) > s]
    t_over = sum(min(e, int(m["End_Timestamp"])) - max(s, int(m["Start_Timestamp"])) for m in over)
    print(f"{name(r):40s} {qcol} {r[qcol]:>4s}  {(e - s) / 1e3:8.1f} us; main-queue kernels running meanwhile: {len(over)} "
          f"({t_over / 1e3:.1f} us of overlap): {', '.join(sorted({name(m) for m in over}))[:150]}")
    shown += 1
    if shown >= 16:
        break
n_side = len(side)
tot_side = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in side)
tot_over = 0
for r in side:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot_over += sum(max(0, min(e, int(m["End_Timestamp"])) - max(s, int(m["Start_Timestamp"]))) for m in tail if m[qcol] == main_q
                    and int(m["Start_Timestamp"]) < e and int(m["End_Timestamp"]) > s)
print(f"side-queue launches {n_side}, their total time {tot_side / 1e6:.3f} ms, of which under main-queue kernels {tot_over / 1e6:.3f} ms")
