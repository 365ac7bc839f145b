#!/bin/bash
# Whole-step HBM traffic (SURVEY 8(d) estimates ~0.7 GB per image at level 5): rocprofv3 PMC over D+G steps of bench.py, FETCH_SIZE and
# WRITE_SIZE in separate passes (MI355X_MICROARCH.md: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2; gfx950 reports half the bytes of a
# wide coalesced read -> x 2), summed over EVERY dispatch of the run; two run lengths per counter, bytes per step = difference / extra
# steps (set-up, warm-up and the dominant-kernel probe cancel).  bash tools/measure_step_traffic.sh [level] [batch]
#   -> gpurun_out/traffic_step_l<level>.json   (copy into profiles/ to have bench.py quote it)
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=${1:-5}; B=${2:-64}
S1=6; S2=16
cd /tmp && export TMPDIR=/tmp
for P in FETCH_SIZE WRITE_SIZE; do
  for S in $S1 $S2; do
    D=$R/gpurun_out/pmcs_${P}_$S
    rm -rf $D
    rocprofv3 --kernel-trace --pmc $P --output-format csv -d $D -- python3 $R/bench.py --level $L --batch $B --steps $S --warmup 4 --no-extra --no-cpu-baseline --no-cadence > /dev/null 2>&1 || exit 1
  done
done
python3 - "$R" "$L" "$B" "$S1" "$S2" <<'PY'
import csv, glob, json, sys
R, L, B, S1, S2 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
def total(P, S):
    f = glob.glob(f"{R}/gpurun_out/pmcs_{P}_{S}/*/*counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == P]
    return sum(float(r["Counter_Value"]) for r in rows), len(rows)
f1, n1 = total("FETCH_SIZE", S1); f2, n2 = total("FETCH_SIZE", S2)
w1, _ = total("WRITE_SIZE", S1); w2, _ = total("WRITE_SIZE", S2)
fetch_kb, write_kb = (f2 - f1) / (S2 - S1), (w2 - w1) / (S2 - S1)
per_step = (2.0 * fetch_kb + write_kb) * 1024.0
rec = {"level": L, "batch": B, "what": "HBM bytes of one D+G step (critic update with penalty + generator update, Adam on both), every kernel",
       "dispatches_per_step": (n2 - n1) / (S2 - S1), "FETCH_SIZE_KB_raw_per_step": fetch_kb, "WRITE_SIZE_KB_per_step": write_kb,
       "fetch_correction": 2.0, "bytes_per_step": per_step, "bytes_per_image": per_step / B,
       "survey_8d_estimate_bytes_per_image": 0.7e9, "over_estimate": per_step / B / 0.7e9,
       "ms_at_8_TB_per_s": per_step / 8e12 * 1e3,
       "method": f"rocprofv3 --pmc, separate passes, runs of {S1} and {S2} steps, difference / {S2 - S1}; tools/measure_step_traffic.sh"}
json.dump(rec, open(f"{R}/gpurun_out/traffic_step_l{L}.json", "w"), indent=1)
print(json.dumps(rec))
PY
rm -rf $R/gpurun_out/pmcs_*
