#!/bin/bash
# HBM traffic of one kernel from rocprofv3 PMC counters, collected in separate passes as /opt/skills/guides/MI355X_MICROARCH.md
# prescribes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced stream
# -> x2).  Run on the GPU box from the repo root:
#   bash tools/measure_traffic.sh          -> gpurun_out/traffic_dominant_kernel.json  (Winograd conv 48->64@128x128, 192 images)
#   bash tools/measure_traffic.sh stft     -> gpurun_out/traffic_stft_kernel.json      (stft1024_kernel, one 10-minute file)
#   bash tools/measure_traffic.sh codec    -> gpurun_out/traffic_codec_kernels.json    (codec_row_pass + codec_normalize_inplace)
# Copy the JSON into profiles/ to have bench.py quote it.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
CASE=${1:-wino3n}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_fetch_$CASE $R/gpurun_out/pmc_write_$CASE
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$CASE -- python3 $R/tools/prof_one.py $CASE 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$CASE -- python3 $R/tools/prof_one.py $CASE 3 > /dev/null 2>&1
python3 - "$R" "$CASE" <<'PY'
import csv, glob, json, sys
R, CASE = sys.argv[1], sys.argv[2]
T = 1 + 44100 * 600 // 256
CASES = {
    "wino3n": ("traffic_dominant_kernel.json", [("wino3x3_strip", "wino3x3_strip<2,8,ACT_POOL_MOUT> lrelu+avgpool+tile mask 48->64@128x128, 192 images (pooled y and mask bytes written; two workgroup rows of 32 out-channels each read the input)",
                192 * (4.0 * 48 * 128 * 128 + 4.0 * 64 * 64 * 64 + 1.0 * 64 * 64 * 64))]),
    "stft": ("traffic_stft_kernel.json", [("stft1024_kernel", "stft1024_kernel, one 10-minute mono 44.1 kHz file (103 360 frames)", 5120.0 * T)]),
    "codec": ("traffic_codec_kernels.json", [("codec_row_pass", "codec_row_pass, 512 x 103 360 bins: 8 B in + 8 B (raw images) out per bin", 16.0 * 512 * T),
                                              ("codec_normalize_inplace", "codec_normalize_inplace: 8 B in + 8 B out per stored bin", 16.0 * 512 * 201 * 512)]),
}
def avg(kind, name, kern):
    f = glob.glob(f"{R}/gpurun_out/pmc_{kind}_{CASE}/*/*counter_collection.csv")[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v)
fname, kernels = CASES[CASE]
recs = []
for kern, what, alg in kernels:
    fetch_kb, write_kb = avg("fetch", "FETCH_SIZE", kern), avg("write", "WRITE_SIZE", kern)
    rec = {"kernel": what, "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb, "fetch_correction": 2.0,
           "bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0, "algorithmic_bytes": alg}
    rec["traffic_over_algorithmic"] = rec["bytes_per_launch"] / alg
    recs.append(rec)
out = recs[0] if len(recs) == 1 else {"kernels": recs, "bytes_per_launch": sum(r["bytes_per_launch"] for r in recs)}
json.dump(out, open(f"{R}/gpurun_out/{fname}", "w"), indent=1)
print(json.dumps(out))
PY
