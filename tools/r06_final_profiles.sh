set -x
cd $GRAFT_REPO_ROOT
TAG=r06 bash tools/profiles.sh "5:64 4:32 3:8 6:6 7:6 7:16" > gpurun_out/r06_profiles_log.txt 2>&1
MG_WGRAD_ROWS=1 bash tools/pmc_wino.sh w20n wino_wgrad_rows > gpurun_out/r06_pmc_w20n_rows.txt 2>&1
MG_WGRAD_ROWS=0 bash tools/pmc_wino.sh w20n wino_wgrad_mfma > gpurun_out/r06_pmc_w20n_chunk.txt 2>&1
bash tools/pmc_wino.sh ww16 wino_wgrad_narrow > gpurun_out/r06_pmc_ww16_narrow.txt 2>&1
for m in 0 1; do MG_WGRAD_ROWS=$m python tools/ab_wgrad_fast.py > gpurun_out/r06_ab_wgrad_rows_$m.txt 2>&1; done
bash tools/ab_levels.sh MG_WGRAD_ROWS "5:64 4:32 6:6 7:6" "0 1 0 1" > gpurun_out/r06_ab_levels_rows_final.txt 2>&1
python tools/l67_bounds.py 7 6 3 40 > gpurun_out/r06_l67_bounds_l7.txt 2>&1
python tools/l67_bounds.py 6 6 3 40 > gpurun_out/r06_l67_bounds_l6.txt 2>&1
for a in 0 2; do MG_WGRAD_ROWS_ABLATE=$a bash tools/pmc_clock.sh w20n wino_wgrad_rows; done > gpurun_out/r06_clock_rows.txt 2>&1
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
tail -c 300 gpurun_out/r06_bench_line.json
