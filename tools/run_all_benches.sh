set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/r02f; mkdir -p $o
timeout -k 10 400 python bench.py > $o/bench_default.json 2> $o/bench_default.err
for gb in 512 256 128; do timeout -k 10 200 python bench.py --global-batch $gb --no-cpu-baseline > $o/bench_cfg2_b$gb.json 2> $o/b.err; done
timeout -k 10 200 python bench.py --config cfg1 --no-cpu-baseline > $o/bench_cfg1.json 2> $o/b.err
timeout -k 10 300 python bench.py --config cfg3 --global-batch 64 --no-cpu-baseline > $o/bench_cfg3.json 2> $o/b.err
timeout -k 10 300 python bench.py --config cfg4 --global-batch 128 --no-cpu-baseline > $o/bench_cfg4.json 2> $o/b.err
for f in $o/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d["value"],1), round(d["ms_per_step"],2), round(d["roofline"]["achieved"],1), round(d["roofline"]["whole_step"]["achieved"],1), d["config"]["hbm_peak_gb"])
PY
done
