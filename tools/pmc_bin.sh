#!/bin/bash
# SQ counters of a stand-alone binary (two --pmc passes): bash tools/pmc_bin.sh TAG KERNEL_SUBSTR -- ./binary args...   (env vars pass through)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; K=$2; shift 3
BIN=$(realpath $1); shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcb_${TAG}_a $R/gpurun_out/pmcb_${TAG}_b
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcb_${TAG}_a -- $BIN "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmcb_${TAG}_b -- $BIN "$@" > /dev/null 2>&1
python3 - "$R" "$TAG" "$K" <<'PY'
import csv, glob, sys, collections
R, C, K = sys.argv[1], sys.argv[2], sys.argv[3]
for k in "ab":
    fs = glob.glob(f"{R}/gpurun_out/pmcb_{C}_{k}/*/*counter_collection.csv")
    if not fs: print("no output for pass", k); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "pack" in r["Kernel_Name"] or K not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, v in acc.items(): print(f"{C:6s} {n:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
rm -rf $R/gpurun_out/pmcb_${TAG}_a $R/gpurun_out/pmcb_${TAG}_b
