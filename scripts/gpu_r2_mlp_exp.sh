for v in "" "TS_NO_SIDE_STREAM=1" "TS_MLP_PER_LAYER=1 TS_NO_SIDE_STREAM=1"; do
  for w in redq sac; do
    env $v python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v]', '$w', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
