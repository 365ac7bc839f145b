# The tracked profiles of round 6 (run on the GPU box from the repo root; results in gpurun_out/, copied into profiles/ afterwards).
set -x
cd $GRAFT_REPO_ROOT
TAG=r06 bash tools/profiles.sh "5:64 4:32 3:8 6:6 7:6 7:16" > gpurun_out/r06_profiles_log.txt 2>&1
MG_WGRAD_ROWS=1 bash tools/pmc_wino.sh w20n wino_wgrad_rows > gpurun_out/r06_pmc_w20n_rows.txt 2>&1
bash tools/pmc_wino.sh up54 upconv3x3 > gpurun_out/r06_pmc_upconv_up54.txt 2>&1
python tools/ab_winoups.py 2>&1 | grep -v amdgpu > gpurun_out/r06_ab_winoups.txt
python tools/l67_bounds.py 7 6 3 40 > gpurun_out/r06_l67_bounds_l7.txt 2>&1
python tools/l67_bounds.py 6 6 3 40 > gpurun_out/r06_l67_bounds_l6.txt 2>&1
python tools/l67_bounds.py 5 64 3 15 > gpurun_out/r06_bounds_l5.txt 2>&1
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
tail -c 300 gpurun_out/r06_bench_line.json
