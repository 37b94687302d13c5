#!/bin/bash
# Round profile run (on the GPU box, through gpurun): kernel-trace summaries of the default bench (cfg2, global batch 1024) and of the per-rank
# batch of the 8-GPU point (128), then the PMC passes - FETCH_SIZE, WRITE_SIZE and the SQ busy / wait counters in SEPARATE runs, the program
# directly after `--` (MI355X_MICROARCH.md, rocprofv3 PMC slots).  Output under gpurun_out/prof_$TAG; tools/db_kernel_stats.py and
# tools/collect_pmc.py reduce it to the files committed under profiles/.
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench default done"
python3 bench.py --steps 20 --warmup 5 --global-batch 128 --no-cpu-baseline > $OUT/bench_b128.json 2> $OUT/bench_b128.err
echo "bench b128 done"
python3 bench.py --config ref_swin --steps 20 --warmup 5 > $OUT/bench_ref_swin.json 2> $OUT/bench_ref_swin.err
echo "bench ref_swin done"
rocprofv3 --kernel-trace --stats -d $OUT/trace1024 -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --path engine > $OUT/trace1024.log 2>&1
echo "trace 1024 done"
rocprofv3 --kernel-trace --stats -d $OUT/trace128 -o t -- python3 bench.py --steps 6 --warmup 2 --global-batch 128 --no-cpu-baseline --path engine > $OUT/trace128.log 2>&1
echo "trace 128 done"
rocprofv3 --kernel-trace --stats -d $OUT/traceswin -o t -- python3 bench.py --config ref_swin --path engine --steps 6 --warmup 2 > $OUT/traceswin.log 2>&1
echo "trace ref_swin done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $OUT/pmc_$C -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --path engine > $OUT/pmc_$C.log 2>&1
  echo "pmc $C done"
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_SQ -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --path engine > $OUT/pmc_SQ.log 2>&1
echo "pmc SQ done"
# reduce on the box: the databases are tens of MB each and gpurun_out/ returns at most 64 MiB
python3 tools/db_kernel_stats.py $OUT/trace1024/*/t_results.db $OUT/kernel_stats_cfg2_gb1024.csv 9 > $OUT/kernel_stats_cfg2_gb1024.txt 2>&1 || python3 tools/db_kernel_stats.py $OUT/trace1024/t_results.db $OUT/kernel_stats_cfg2_gb1024.csv 9 > $OUT/kernel_stats_cfg2_gb1024.txt 2>&1
python3 tools/db_kernel_stats.py $OUT/trace128/*/t_results.db $OUT/kernel_stats_cfg2_b128.csv 9 > $OUT/kernel_stats_cfg2_b128.txt 2>&1 || python3 tools/db_kernel_stats.py $OUT/trace128/t_results.db $OUT/kernel_stats_cfg2_b128.csv 9 > $OUT/kernel_stats_cfg2_b128.txt 2>&1
python3 tools/db_kernel_stats.py $(ls $OUT/traceswin/t_results.db $OUT/traceswin/*/t_results.db 2>/dev/null | head -1) $OUT/kernel_stats_ref_swin_b32.csv 8 > $OUT/kernel_stats_ref_swin_b32.txt 2>&1
python3 tools/db_timeline.py $(ls $OUT/trace1024/t_results.db $OUT/trace1024/*/t_results.db 2>/dev/null | head -1) > $OUT/timeline_cfg2_gb1024.txt 2>&1
python3 tools/db_timeline.py $(ls $OUT/trace128/t_results.db $OUT/trace128/*/t_results.db 2>/dev/null | head -1) > $OUT/timeline_cfg2_b128.txt 2>&1
F=$(ls $OUT/pmc_FETCH_SIZE/p_results.db $OUT/pmc_FETCH_SIZE/*/p_results.db 2>/dev/null | head -1)
W=$(ls $OUT/pmc_WRITE_SIZE/p_results.db $OUT/pmc_WRITE_SIZE/*/p_results.db 2>/dev/null | head -1)
S=$(ls $OUT/pmc_SQ/p_results.db $OUT/pmc_SQ/*/p_results.db 2>/dev/null | head -1)
python3 tools/collect_pmc.py $OUT/pmc_cfg2_gb1024.json fetch=$F write=$W sq=$S > $OUT/pmc_cfg2_gb1024.txt 2>&1
rm -rf $OUT/trace1024 $OUT/trace128 $OUT/traceswin $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ
for f in $OUT/*.log; do tail -n 2 $f; done
ls -la $OUT
