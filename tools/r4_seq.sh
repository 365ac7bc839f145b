#!/bin/bash
# Sequence traces (one steady-state step in launch order) of the small configurations.  bash tools/r4_seq.sh "3:8 4:32 6:6"
R=${GRAFT_REPO_ROOT:-$(pwd)}
CASES=${1:-"3:8 4:32 6:6"}
TAG=${2:-r4}
cd /tmp && export TMPDIR=/tmp
for c in $CASES; do
  L=${c%%:*}; B=${c##*:}
  python3 $R/bench.py --level $L --batch $B --steps 200 --warmup 100 --no-extra --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_l${L}_bs${B}.json 2> $R/gpurun_out/${TAG}_bench_l${L}_bs${B}.err || exit 1
  rm -rf $R/gpurun_out/${TAG}_trace_l${L}_bs${B}
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_trace_l${L}_bs${B} -- python3 $R/bench.py --level $L --batch $B --steps 120 --warmup 40 --no-extra --no-cpu-baseline > $R/gpurun_out/${TAG}_prof_l${L}_bs${B}.json 2>/dev/null || exit 1
  f=$(ls $R/gpurun_out/${TAG}_trace_l${L}_bs${B}/*/*kernel_trace.csv | head -1)
  python3 $R/tools/trace_seq.py $f 2 > $R/gpurun_out/${TAG}_seq_l${L}_bs${B}.txt
  python3 $R/tools/trace_table.py $f 60 > $R/gpurun_out/${TAG}_table_l${L}_bs${B}.txt
  rm -rf $R/gpurun_out/${TAG}_trace_l${L}_bs${B}
  echo "== L$L bs$B"; python3 -c "import json;d=json.load(open('$R/gpurun_out/${TAG}_bench_l${L}_bs${B}.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['executed_frac'])"
  head -1 $R/gpurun_out/${TAG}_table_l${L}_bs${B}.txt
done
