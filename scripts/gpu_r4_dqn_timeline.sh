#!/bin/bash
# per-dispatch timeline of one C3 DQN update (all streams as in production) -> gpurun_out/r4tl/dqn_timeline.txt
O=$GRAFT_REPO_ROOT/gpurun_out/r4tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof -o dqn -- python $GRAFT_REPO_ROOT/bench_dqn.py --steps 12 --warmup 5 --no-cpu-baseline > $O/dqn_tl.json 2> $O/dqn_tl.err
cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/schema.txt 2>&1
import sqlite3,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4tl/prof/**/*.db",recursive=True)
print(f)
db=sqlite3.connect(f[0])
for r in db.execute("select name, sql from sqlite_master where name like '%kernel%' limit 6"): print(r[0], (r[1] or '')[:1500])
PY
DB=$(find $O/prof -name '*.db' | head -1)
python scripts/rocprof_timeline.py $DB "adam_kernel(" 14 > $O/dqn_timeline.txt 2>&1
rm -rf $O/prof
tail -80 $O/dqn_timeline.txt
