#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r6h; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" "TS_WGRAD_TILE64=1"; do
  for wl in sac td3 ddpg redq dsac; do
    env $v timeout 300 python bench.py --workload $wl --no-cpu-baseline > $O/tmp.json 2>> $O/err.txt
    python - <<PY
import json
d = json.loads(open("$O/tmp.json").read().strip().splitlines()[-1])
r = d.get("roofline") or {}
k = r.get("kernel_us_per_update") or {kk: round(v["us_per_update"], 1) for kk, v in (d.get("roofline_by_kind") or {}).items()}
print("$wl [$v]", round(d["value"], 1), d.get("unit"), "frac", round(r.get("frac") or 0, 4), k)
PY
  done
done
done
grep -v amdgpu.ids $O/err.txt | tail -5
