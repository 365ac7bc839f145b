#!/bin/bash
# Tracked profiles of a round (run on the GPU box from the repo root; results in gpurun_out/, copy what is to be judged into profiles/):
#   bash tools/profiles.sh "5:64 4:32 3:8 6:6 7:6 7:16" [steps-only]      (TAG=r05 by default: the prefix of every file)
#   per configuration: rocprofv3 kernel stats + steady-state per-kernel table of a 160-step run whose own ms_per_step is printed
#   beside the trace's busy time; the dominant launch alone (stats, clock, SQ counters, HBM traffic); the audio kernels.
R=${GRAFT_REPO_ROOT:-$(pwd)}
CASES=${1:-"5:64 4:32 3:8 6:6 7:6 7:16"}
TAG=${TAG:-r05}
cd /tmp && export TMPDIR=/tmp
for c in $CASES; do
  L=${c%%:*}; B=${c##*:}
  D=$R/gpurun_out/${TAG}_trace_l${L}_bs${B}
  rm -rf $D
  python3 $R/bench.py --level $L --batch $B --steps 160 --warmup 40 --no-extra --no-cpu-baseline --no-cadence > $R/gpurun_out/${TAG}_unprofiled_run_l${L}_bs${B}.json 2>/dev/null || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --level $L --batch $B --steps 160 --warmup 40 --no-extra --no-cpu-baseline --no-cadence > $R/gpurun_out/${TAG}_profiled_run_l${L}_bs${B}.json 2>/dev/null || exit 1
  cp $D/*/*kernel_stats.csv $R/gpurun_out/${TAG}_bench_l${L}_bs${B}_kernel_stats.csv
  f=$(ls $D/*/*kernel_trace.csv | head -1)
  python3 $R/tools/trace_table.py $f 45 $R/gpurun_out/${TAG}_profiled_run_l${L}_bs${B}.json $R/gpurun_out/${TAG}_unprofiled_run_l${L}_bs${B}.json > $R/gpurun_out/${TAG}_trace_table_l${L}_bs${B}.txt
  head -1 $R/gpurun_out/${TAG}_trace_table_l${L}_bs${B}.txt
  rm -rf $D
done
if [ "$2" != "steps-only" ]; then
  rm -rf $R/gpurun_out/dom
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dom -- python3 $R/tools/prof_one.py wino3n 300 > /dev/null 2>&1 || exit 1
  cp $R/gpurun_out/dom/*/*kernel_stats.csv $R/gpurun_out/${TAG}_dominant_kernel_wino3n_stats.csv
  rm -rf $R/gpurun_out/dom
  cd $R && bash tools/pmc_wino.sh wino3n wino3x3_strip > gpurun_out/${TAG}_pmc_sq_counters_wino3n.txt 2>&1
  bash tools/measure_traffic.sh wino3n | tail -1
  bash tools/measure_traffic.sh stft | tail -1
  bash tools/measure_traffic.sh codec | tail -1
  bash tools/measure_step_traffic.sh 5 64 | tail -1
  cd /tmp
  rm -rf $R/gpurun_out/aud
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/aud -- python3 $R/tools/prof_audio.py 300 > /dev/null 2>&1 || exit 1
  cp $R/gpurun_out/aud/*/*kernel_stats.csv $R/gpurun_out/${TAG}_audio_kernel_stats.csv
  rm -rf $R/gpurun_out/aud $R/gpurun_out/sto
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sto -- python3 $R/tools/prof_one.py stft 400 > /dev/null 2>&1 || exit 1
  cp $R/gpurun_out/sto/*/*kernel_stats.csv $R/gpurun_out/${TAG}_stft_only_kernel_stats.csv
  rm -rf $R/gpurun_out/sto $R/gpurun_out/pmc_*
fi
# data-parallel overlap: one rank, RCCL, MG_FORCE_DP=1 -- which queue the all-reduce + Adam run on and what the main queue does meanwhile
if [ "$2" != "steps-only" ]; then
  cd /tmp
  rm -rf $R/gpurun_out/dpt
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 MG_FORCE_DP=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/dpt -- python3 $R/bench.py --gpus 1 --level 5 --batch 64 --steps 40 --warmup 20 --no-extra --no-cpu-baseline --no-cadence > $R/gpurun_out/${TAG}_dp_profiled_run.json 2>$R/gpurun_out/${TAG}_dp_profiled_run.err
  f=$(ls $R/gpurun_out/dpt/*/*kernel_trace.csv | head -1)
  python3 $R/tools/dp_overlap.py $f > $R/gpurun_out/${TAG}_dp_overlap_1rank_nccl.txt 2>&1
  tail -3 $R/gpurun_out/${TAG}_dp_overlap_1rank_nccl.txt
  rm -rf $R/gpurun_out/dpt
fi
