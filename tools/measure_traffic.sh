#!/bin/bash
# HBM traffic of the dominant kernel (Winograd conv 48->64@128x128 + lrelu + avgpool + tile mask over 3x64 images) from rocprofv3 PMC counters, collected in
# separate passes as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2; on gfx950
# FETCH_SIZE reports half the bytes of a wide coalesced stream -> x2).  Run on the GPU box from the repo root:
#   bash tools/measure_traffic.sh        -> profiles/traffic_dominant_kernel.json
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/prof_one.py wino3n 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/prof_one.py wino3n 3 > /dev/null 2>&1
python3 - "$R" <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
def avg(kind, name):
    f = glob.glob(f"{R}/gpurun_out/pmc_{kind}/*/*counter_collection.csv")[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "wino3x3_mfma" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v)
fetch_kb, write_kb = avg("fetch", "FETCH_SIZE"), avg("write", "WRITE_SIZE")
out = {"kernel": "wino3x3_mfma<2,2,4> lrelu+avgpool+tile mask 48->64@128x128, 192 images (pooled y and mask bytes written)",
       "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb, "fetch_correction": 2.0,
       "bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
       "algorithmic_bytes": 192 * (4.0 * 48 * 128 * 128 + 4.0 * 64 * 64 * 64 + 1.0 * 64 * 64 * 64)}
out["traffic_over_algorithmic"] = out["bytes_per_launch"] / out["algorithmic_bytes"]
json.dump(out, open(f"{R}/gpurun_out/traffic_dominant_kernel.json", "w"), indent=1)
print(json.dumps(out))
PY
