cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MEDMOE_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b128s -o b128s -- python3 bench.py --steps 6 --warmup 3 --global-batch 128 --no-cpu-baseline --path engine > gpurun_out/prof_b128s.log 2>&1
MEDMOE_OVERLAP_WGRAD=0 python bench.py --steps 20 --warmup 5 --global-batch 128 --no-cpu-baseline --path engine > gpurun_out/bench_b128s.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_b128s.log | head -1
