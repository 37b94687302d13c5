cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for c in cfg4 cfg4_bf16; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$c -o t -- python3 bench.py --config $c --global-batch 128 --steps 4 --warmup 2 --no-cpu-baseline --path engine > gpurun_out/prof_$c.log 2>&1
  python3 tools/db_kernel_stats.py $(ls gpurun_out/prof_$c/t_results.db gpurun_out/prof_$c/*/t_results.db 2>/dev/null | head -1) gpurun_out/r03_${c}_b128_kernel_stats.csv 7 > gpurun_out/r03_${c}_b128_kernel_stats.txt
  rm -rf gpurun_out/prof_$c
done
