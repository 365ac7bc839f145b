#!/bin/bash
# Memory-side counters of one kernel (separate --pmc passes): bash tools/pmc_mem.sh [prof_one case] [kernel-name substring]
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=${1:-dg16}
K=${2:-wino3x3}
cd /tmp && export TMPDIR=/tmp
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  T=$(echo $P | tr ' ' '_')
  rm -rf $R/gpurun_out/pmcm_$T
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmcm_$T -- python3 $R/tools/prof_one.py $C 3 > /dev/null 2>&1
  python3 - "$R" "$T" "$K" <<'PY'
import csv, glob, sys, collections
R, T, K = sys.argv[1:4]
fs = glob.glob(f"{R}/gpurun_out/pmcm_{T}/*/*counter_collection.csv")
if not fs:
    print("no output for", T)
else:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if K in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, v in acc.items():
        print(f"{n:28s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
  rm -rf $R/gpurun_out/pmcm_$T
done
