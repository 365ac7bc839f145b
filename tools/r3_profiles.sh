#!/bin/bash
# round-3 tracked profiles: rocprofv3 kernel stats of the bench command per level, the dominant launch alone, its SQ counters and
# effective clock, HBM traffic.  bash tools/r3_profiles.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in 5:64 4:32 6:6 7:6 7:16; do
  L=${c%%:*}; B=${c##*:}
  rm -rf $R/gpurun_out/ks_${L}_${B}
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_${L}_${B} -- python3 $R/bench.py --level $L --batch $B --steps 40 --warmup 10 --no-extra --no-cpu-baseline > $R/gpurun_out/ks_${L}_${B}.json 2>/dev/null || exit 1
  cp $R/gpurun_out/ks_${L}_${B}/*/*kernel_stats.csv $R/gpurun_out/r03_bench_l${L}_bs${B}_kernel_stats.csv
  rm -rf $R/gpurun_out/ks_${L}_${B}
done
rm -rf $R/gpurun_out/dom
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dom -- python3 $R/tools/prof_one.py wino3n 300 > /dev/null 2>&1 || exit 1
cp $R/gpurun_out/dom/*/*kernel_stats.csv $R/gpurun_out/r03_dominant_kernel_wino3n_stats.csv
rm -rf $R/gpurun_out/dom $R/gpurun_out/clk
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/clk -- python3 $R/tools/prof_one.py wino3n 300 > /dev/null 2>&1 || exit 1
python3 - "$R" <<'PY'
import csv, glob, sys
R = sys.argv[1]
f = glob.glob(f"{R}/gpurun_out/clk/*/*counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "wino3x3_mfma" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
kt = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(glob.glob(f"{R}/gpurun_out/clk/*/*kernel_trace.csv")[0]))}
vals = [(float(r["Counter_Value"]), kt.get(r["Dispatch_Id"])) for r in rows if kt.get(r["Dispatch_Id"])]
vals = vals[len(vals) // 3:]  # past the clock ramp
clk = [v / 8.0 / (ns * 1e-9) / 1e9 for v, ns in vals]
open(f"{R}/gpurun_out/r03_dominant_kernel_clock.txt", "w").write(
    f"wino3x3_mfma<2,2,4> 48->64@128x128 x192 images: GRBM_GUI_ACTIVE / 8 / kernel wall time over the last {len(clk)} of 300 launches: "
    f"mean {sum(clk) / len(clk):.3f} GHz (min {min(clk):.3f}, max {max(clk):.3f}); mean duration {sum(ns for _, ns in vals) / len(vals) / 1e3:.1f} us (profiled run)\n")
print(open(f"{R}/gpurun_out/r03_dominant_kernel_clock.txt").read())
PY
rm -rf $R/gpurun_out/clk
cd $R && bash tools/pmc_wino.sh wino3n wino3x3_mfma > gpurun_out/r03_pmc_sq_counters_wino3n.txt 2>&1; tail -17 gpurun_out/r03_pmc_sq_counters_wino3n.txt
bash tools/measure_traffic.sh wino3n | tail -1
rm -rf gpurun_out/pmc_*
