cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_swin -o t -- python3 bench.py --config ref_swin --path engine --steps 6 --warmup 2 > gpurun_out/prof_swin.log 2>&1
DB=$(ls gpurun_out/prof_swin/t_results.db gpurun_out/prof_swin/*/t_results.db 2>/dev/null | head -1)
python3 tools/db_kernel_stats.py $DB gpurun_out/swin_b32_kernel_stats.csv 8 | head -70
rm -rf gpurun_out/prof_swin
