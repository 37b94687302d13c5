cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b128 -o b128 -- python3 bench.py --steps 6 --warmup 3 --global-batch 128 --no-cpu-baseline > gpurun_out/prof_b128.log 2>&1
python bench.py --steps 20 --warmup 5 --global-batch 128 --no-cpu-baseline > gpurun_out/bench_b128.log 2>&1
tail -2 gpurun_out/bench_b128.log
ls gpurun_out/prof_b128 | head
