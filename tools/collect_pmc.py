#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc passes (rocpd databases, one per pass) to per-kernel totals: usage
  collect_pmc.py out.json label=path/to/results.db [label=...]
Each database holds one pass (FETCH_SIZE | WRITE_SIZE | the SQ set: the TCC counters do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC
slots").  Output: {kernel name: {counter: sum over the kernel's dispatches, "dispatches": n, "duration_ns": sum}} for every kernel above
0.2 % of the pass's kernel time, plus the corrected HBM bytes where both TCC passes are present: FETCH_SIZE and WRITE_SIZE are in KB;
FETCH_SIZE counts half of the bytes of wide coalesced reads on gfx950 (x 2, MI355X_MICROARCH.md "HBM"), WRITE_SIZE is exact for 16-byte stores."""
import json
import sqlite3
import sys

out_path, passes = sys.argv[1], dict(a.split("=", 1) for a in sys.argv[2:])
res = {}
for label, path in passes.items():
    db = sqlite3.connect(path)
    rows = db.execute("select K.name, P.counter_name, sum(P.counter_value), count(distinct P.dispatch_id), sum(K.end - K.start) "
                      "from pmc_events P join kernels K on K.dispatch_id = P.dispatch_id group by K.name, P.counter_name").fetchall()
    tot = sum(r[4] for r in rows if r[1] == rows[0][1]) or 1
    for name, counter, value, n, dur in rows:
        e = res.setdefault(name, {})
        e[counter] = value
        e.setdefault("dispatches", n)
        e[f"duration_ns_{label}_pass"] = dur
        e.setdefault("share_of_pass", dur / tot)
res = {k: v for k, v in res.items() if v.get("share_of_pass", 0) >= 0.002}
for name, e in res.items():
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        n = max(1, e["dispatches"])
        e["hbm_read_bytes_per_launch_x2_corrected"] = 2.0 * e["FETCH_SIZE"] * 1024 / n
        e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE"] * 1024 / n
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch_x2_corrected"] + e["hbm_write_bytes_per_launch"]
    if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
        wc = e["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in e:
                e[c + "_per_wave_cycle"] = e[c] / wc
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; the matrix-pipe busy cycles over the 1024 SIMDs of the chip
        active = e["GRBM_GUI_ACTIVE"] / 8.0
        e["mfma_busy_fraction"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * active)
    elif "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_BUSY_CYCLES"):
        e["mfma_busy_fraction"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * e["SQ_BUSY_CYCLES"])      # SQ_BUSY_CYCLES: per shader engine (32 SIMDs each)
json.dump({"note": "sums over the dispatches of one bench.py --steps 1 --warmup 1 run per pass (so 2 steps + set-up); SQ_WAVE_CYCLES / SQ_WAIT_* / "
                   "SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES cycles (MI355X_MICROARCH.md)", "kernels": res},
          open(out_path, "w"), indent=1)
for name, e in sorted(res.items(), key=lambda kv: -kv[1].get("share_of_pass", 0))[:14]:
    print(f"{100 * e.get('share_of_pass', 0):5.1f} %  {name[:70]:70s} " + " ".join(f"{k}={v:.3g}" for k, v in e.items() if k.endswith("per_launch") or k.endswith("_cycle") or k == "mfma_busy_fraction"))
