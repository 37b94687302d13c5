cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03_all
python bench.py --config cfg1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_all/cfg1.json 2>/dev/null; echo cfg1 done
python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_all/cfg3.json 2>/dev/null; echo cfg3 done
python bench.py --config cfg4 --global-batch 128 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_all/cfg4.json 2>/dev/null; echo cfg4 done
python bench.py --config cfg4_bf16 --global-batch 128 --steps 10 --warmup 3 --no-cpu-baseline --path engine > gpurun_out/r03_all/cfg4_bf16.json 2>/dev/null; echo cfg4_bf16 done
python bench.py --global-batch 512 --steps 10 --warmup 3 --no-cpu-baseline --path engine > gpurun_out/r03_all/cfg2_b512.json 2>/dev/null
python bench.py --global-batch 256 --steps 10 --warmup 3 --no-cpu-baseline --path engine > gpurun_out/r03_all/cfg2_b256.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03_all/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f.split("/")[-1], round(d["ms_per_step"], 2), round(d["value"], 1), round(d["roofline"]["whole_step"]["frac"], 4), round(d["roofline"]["frac"], 4), d.get("module_path", {}).get("ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
