#!/bin/bash
mkdir -p gpurun_out/r05_run7
timeout 300 python scripts/bneck_tile_time.py f16 256 > gpurun_out/r05_run7/time.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r05_run7/pmc -o p -- python3 $GRAFT_REPO_ROOT/scripts/bneck_tile_time.py f16 256 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' >> gpurun_out/r05_run7/time.txt 2>&1
import sqlite3, glob
for db in glob.glob('gpurun_out/r05_run7/pmc/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if t.startswith('rocpd_pmc_event')][0]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    pi = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
    q = f"select s.kernel_name, i.name, count(*), avg(e.value) from {pmc} e join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id join {pi} i on e.pmc_id = i.id where s.kernel_name like '%bneck_tile%' group by 1,2"
    for r in c.execute(q): print(r)
PY
cat gpurun_out/r05_run7/time.txt
