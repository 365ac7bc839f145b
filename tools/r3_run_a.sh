#!/bin/bash
# round-3 check run: GPU tests, bench line, PMC traffic of the audio kernels, rocprof summaries of the audio path
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gputests.log 2>&1; rc=$?
tail -5 gpurun_out/r3_gputests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r3_bench_line.json 2> gpurun_out/r3_bench_line.err || { tail -20 gpurun_out/r3_bench_line.err; exit 1; }
python3 -c "
import json;d=json.load(open('gpurun_out/r3_bench_line.json'))
print('L5', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['executed_frac'])
for k in ('l4_bs32','l6_bs6','l7_bs6','l7_bs16','l3_bs8'): print(k, d[k]['value'], d[k]['ms_per_step'], d[k]['roofline']['frac'])
print('stft', d['stft']['ms_per_file'], d['stft']['roofline']['frac'], d['stft']['stft_plus_codec'], d['stft']['codec'])
print(d['create_dataset_e2e']); print(d['train_loop']); print(d.get('secondary')); print(d['cpu_baseline'])
"
bash tools/measure_traffic.sh stft && bash tools/measure_traffic.sh codec || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stft_only $R/gpurun_out/prof_audio
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stft_only -- python3 $R/tools/prof_one.py stft 400 > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_audio -- python3 $R/tools/prof_audio.py 300 > /dev/null 2>&1 || exit 1
cp $R/gpurun_out/prof_stft_only/*/*kernel_stats.csv $R/gpurun_out/r03_stft_only_kernel_stats.csv
cp $R/gpurun_out/prof_audio/*/*kernel_stats.csv $R/gpurun_out/r03_audio_kernel_stats.csv
rm -rf $R/gpurun_out/prof_stft_only $R/gpurun_out/prof_audio $R/gpurun_out/pmc_*
head -8 $R/gpurun_out/r03_audio_kernel_stats.csv | cut -c1-150; head -3 $R/gpurun_out/r03_stft_only_kernel_stats.csv | cut -c1-150
