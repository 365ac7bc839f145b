#!/bin/bash
# SQ counters of one kernel (two --pmc passes, no other tracing).  bash tools/pmc_wino.sh [prof_one case] [kernel-name substring]
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=${1:-wino}
K=${2:-mfma}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_${C}_a $R/gpurun_out/pmc_${C}_b
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_${C}_a -- python3 $R/tools/prof_one.py $C 3 > $R/gpurun_out/pmc_${C}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc_${C}_b -- python3 $R/tools/prof_one.py $C 3 > $R/gpurun_out/pmc_${C}_b.log 2>&1
python3 - "$R" "$C" "$K" <<'PY'
import csv, glob, sys, collections
R, C, K = sys.argv[1], sys.argv[2], sys.argv[3]
for k in "ab":
    fs = glob.glob(f"{R}/gpurun_out/pmc_{C}_{k}/*/*counter_collection.csv")
    if not fs: print("no output for pass", k); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "pack" in r["Kernel_Name"] or K not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, v in acc.items(): print(f"{n:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
