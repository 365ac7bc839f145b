#!/bin/bash
# Effective shader clock of one kernel: GRBM_GUI_ACTIVE cycles / traced duration.  bash tools/pmc_clock.sh <prof_one case> <kernel substring>
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=${1:-w20n}
K=${2:-wino_wgrad}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_clk
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_clk -- python3 $R/tools/prof_one.py $C 5 > $R/gpurun_out/pmc_clk.log 2>&1
python3 - "$R" "$K" <<'PY'
import csv, glob, sys
R, K = sys.argv[1], sys.argv[2]
cc = glob.glob(f"{R}/gpurun_out/pmc_clk/*/*counter_collection.csv")[0]
kt = glob.glob(f"{R}/gpurun_out/pmc_clk/*/*kernel_trace.csv")[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
for r in csv.DictReader(open(cc)):
    if K in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        d = dur[r["Dispatch_Id"]][0]
        print(f'{r["Kernel_Name"][:60]:60s} {d/1e3:9.1f} us  GUI_ACTIVE {float(r["Counter_Value"]):14.0f}  -> {float(r["Counter_Value"])/d:6.3f} GHz (if the counter is one instance)')
PY
