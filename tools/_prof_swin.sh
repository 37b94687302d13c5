cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_swin -o t -- python3 bench.py --config ref_swin --steps 6 --warmup 2 > gpurun_out/prof_swin.log 2>&1
DB=$(ls gpurun_out/prof_swin/t_results.db gpurun_out/prof_swin/*/t_results.db 2>/dev/null | head -1)
python3 tools/db_kernel_stats.py $DB gpurun_out/swin_b32_kernel_stats.csv 8 | head -45
python3 - <<PY
import sqlite3
db = sqlite3.connect("$DB")
rows = db.execute("select start, end, name from kernels order by start").fetchall()
# last full step: between the last two multi_tensor / fused adam kernels
ad = [i for i, r in enumerate(rows) if "adam" in r[2].lower()]
print("adam launches", len(ad))
if len(ad) >= 3:
    a, b = ad[-3], ad[-2]
    seg = rows[a + 1:b + 1]
    t0, t1 = seg[0][0], seg[-1][1]
    busy = sum(e - s for s, e, n in seg)
    print(f"step wall {(t1 - t0) / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, launches {len(seg)}")
PY
rm -rf gpurun_out/prof_swin
