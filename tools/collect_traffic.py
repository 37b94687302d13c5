#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two separate runs of bench.py) to HBM bytes per launch of the
dominant GEMM kernels, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section): both counters are in KB,
FETCH_SIZE reports half of the bytes of wide coalesced reads (x2), WRITE_SIZE is exact for 16-B-per-lane stores.

  python tools/collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <key> [kernel-substring]
writes/updates profiles/<MEDMOE_TRAFFIC_FILE, default r02_traffic.json>[<key>]."""
import csv, json, os, sys, collections
fetch_csv, write_csv, key = sys.argv[1:4]
pat = sys.argv[4] if len(sys.argv) > 4 else "gemm_nt512_kernel"
def total(path, counter):
    tot, disp = 0.0, set()
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    return tot, len(disp)
f, nf = total(fetch_csv, "FETCH_SIZE")
w, nw = total(write_csv, "WRITE_SIZE")
out = {"kernel": pat, "launches_profiled": nf,
       "fetch_bytes_per_launch_x2_corrected": 2.0 * f * 1024 / max(nf, 1),
       "write_bytes_per_launch": w * 1024 / max(nw, 1)}
out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch_x2_corrected"] + out["write_bytes_per_launch"]
out["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 1 --warmup 1; counters are KB; "
               "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B)")
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", os.environ.get("MEDMOE_TRAFFIC_FILE", "r02_traffic.json"))
d = json.load(open(path)) if os.path.exists(path) else {}
d[key] = out
json.dump(d, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
