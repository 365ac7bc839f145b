#!/bin/bash
# Step time + per-kernel trace of the levels a real run lives at (reference batch 6, train.py:43).  bash tools/prof_levels.sh "7:6 6:6 7:16"
R=${GRAFT_REPO_ROOT:-$(pwd)}
CASES=${1:-"7:6 6:6 7:16"}
cd /tmp && export TMPDIR=/tmp
for c in $CASES; do
  L=${c%%:*}; B=${c##*:}
  python3 $R/bench.py --level $L --batch $B --steps 30 --warmup 10 --no-extra --no-cpu-baseline > $R/gpurun_out/r3_bench_l${L}_bs${B}.json 2> $R/gpurun_out/r3_bench_l${L}_bs${B}.err || exit 1
  rm -rf $R/gpurun_out/r3_trace_l${L}_bs${B}
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3_trace_l${L}_bs${B} -- python3 $R/bench.py --level $L --batch $B --steps 20 --warmup 10 --no-extra --no-cpu-baseline > /dev/null 2>&1 || exit 1
  f=$(ls $R/gpurun_out/r3_trace_l${L}_bs${B}/*/*kernel_trace.csv | head -1)
  python3 $R/tools/trace_table.py $f 45 > $R/gpurun_out/r3_trace_table_l${L}_bs${B}.txt
  rm -rf $R/gpurun_out/r3_trace_l${L}_bs${B}
  echo "== L$L bs$B"; python3 -c "import json;d=json.load(open('$R/gpurun_out/r3_bench_l${L}_bs${B}.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['executed_frac'])"
done
