cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_swin -o t -- python3 bench.py --config ref_swin --path engine --steps 6 --warmup 2 > gpurun_out/prof_swin.log 2>&1
DB=$(ls gpurun_out/prof_swin/t_results.db gpurun_out/prof_swin/*/t_results.db 2>/dev/null | head -1)
python3 - <<PY
import sqlite3
db = sqlite3.connect("$DB")
rows = db.execute("select start, end, name from kernels order by start").fetchall()
n = len(rows)
lo, hi = int(n * 0.62), int(n * 0.88)
big = [i for i in range(lo, hi) if rows[i + 1][0] - rows[i][1] > 150e3]
for i in big:
    print(f"---- gap {(rows[i + 1][0] - rows[i][1]) / 1e3:.0f} us at launch {i}")
    for j in range(i - 5, i + 6):
        mark = " <gap>" if j == i else ""
        print(f"   {(rows[j][1] - rows[j][0]) / 1e3:8.1f} us  {rows[j][2][:110]}{mark}")
PY
rm -rf gpurun_out/prof_swin
