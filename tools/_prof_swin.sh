cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python3 tools/bench_mirror.py 32 10 swin_t 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_swin -o t -- python3 tools/bench_mirror.py 32 6 swin_t > gpurun_out/prof_swin.log 2>&1
python3 tools/db_kernel_stats.py gpurun_out/prof_swin/*/t_results.db gpurun_out/swin_b32_kernel_stats.csv 8 2>/dev/null || python3 tools/db_kernel_stats.py gpurun_out/prof_swin/t_results.db gpurun_out/swin_b32_kernel_stats.csv 8
rm -rf gpurun_out/prof_swin
