#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_dcn.py -x -q -k "one_pass" 2>&1 | tail -5 > gpurun_out/r03_dcn_tests.txt
{
for split in 0 1; do for segs in 0 1 2 3; do
  echo "== split=$split segs=$segs"; RTP_DCN_SPLIT_GW=$split RTP_DCN_SEGS=$segs timeout 300 python3 tools/bench_dcn.py 2>/dev/null | head -1
done; done
} > gpurun_out/r03_dcn_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_dcn -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_dcn.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/r03_dcn_tests.txt gpurun_out/r03_dcn_bench.txt
python3 - <<'PY'
import sqlite3,glob
for d in ("prof_dcn",):
    db=glob.glob('gpurun_out/%s/*.db'%d)[0]
    c=sqlite3.connect(db)
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd=[t for t in tabs if 'kernel_dispatch' in t][0]
    ks=[t for t in tabs if 'kernel_symbol' in t][0]
    q=f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%dcn%' group by s.kernel_name order by 3 desc limit 8"
    print(d)
    for r in c.execute(q): print("  %-80s %5d %10.1f us"%(r[0][:80],r[1],r[2]))
PY
