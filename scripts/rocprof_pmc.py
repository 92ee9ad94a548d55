"""Mean PMC counter values per launch, grouped by kernel name + grid, from rocprofv3 rocpd databases:
    python scripts/rocprof_pmc.py a_results.db [b_results.db ...] [--match conv] [--json out.json]
--json also writes {kernel name: {"grid": [x, y, z], "launches": n, counter: mean per launch, ...,
"by_grid": {"x,y,z": {"launches": n, counter: mean, ...}, ...}}}: the top-level fields are those of the heaviest grid of the
name, `by_grid` keeps EVERY launch geometry apart (round 5's file had the 2^24 launches of `gae_single_pass` overwrite the 2^20
ones, so the bench line quoted a 16x over-fetch that does not exist) -- bench.py reads profiles/r06_pmc_hbm_traffic.json,
produced this way, for its `roofline.traffic` fields and looks its kernels up by (name, grid)."""
import collections
import json
import sqlite3
import sys

match = None
json_out = None
dbs = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--match":
        match = args.pop(0)
    elif a == "--json":
        json_out = args.pop(0)
    else:
        dbs.append(a)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in dbs:
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if view is None:
        print("no counters view in", path, [t for t in tabs if "count" in t.lower() or "pmc" in t.lower()])
        continue
    cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
    q = f"select kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name, value from {view}" \
        if "grid_size_x" in cols else None
    if q is None:
        print(cols)
        continue
    for name, gx, gy, gz, cname, val in db.execute(q):
        if match and match not in name:
            continue
        key = (name.replace("(anonymous namespace)::", "").replace("void ", "")[:48], gx, gy, gz)
        e = acc[key][cname]
        e[0] += val
        e[1] += 1
for key in sorted(acc):
    c = {k: v[0] / v[1] for k, v in acc[key].items()}
    n = next(iter(acc[key].values()))[1]
    print(key, "launches", n)
    print("   ", {k: round(v) for k, v in sorted(c.items())})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        print("    mfma_busy/ (busy_cycles*4 simd-ish):", round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c["SQ_BUSY_CYCLES"], 1), 3),
              " wait_any/wave_cycles:", round(c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 3),
              " wait_inst/wave_cycles:", round(c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 3))

if json_out:
    best, grids = {}, collections.defaultdict(dict)
    for key in acc:
        name = key[0].split("<")[0].split("(")[0].strip().split("::")[-1]
        if name == "conv_rows_kernel" and "<" in key[0]:      # <false, ...> forward / <true, ...> input gradient: two bench kinds
            name += "<" + key[0].split("<", 1)[1].split(",")[0].strip()
        c = {k: v[0] / v[1] for k, v in acc[key].items()}
        n = next(iter(acc[key].values()))[1]
        weight = n * key[1] * max(key[2], 1) * max(key[3], 1)
        entry = {"launches": n, **{k: v for k, v in sorted(c.items())}}
        gkey = ",".join(str(int(x)) for x in key[1:])
        if gkey in grids[name]:                               # template instantiations of one name with the same grid: pool them
            old = grids[name][gkey]
            tot = old["launches"] + n
            entry = {"launches": tot, **{k: (old.get(k, v) * old["launches"] + v * n) / tot for k, v in sorted(c.items())}}
        grids[name][gkey] = entry
        if name not in best or weight > best[name][0]:
            best[name] = (weight, {"grid": list(key[1:]), **entry})
    with open(json_out, "w") as f:
        json.dump({k: {**v[1], "by_grid": grids[k]} for k, v in sorted(best.items())}, f, indent=1)
