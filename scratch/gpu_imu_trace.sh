#!/bin/bash
# durations of the consecutive k_imu launches of small visual-inertial windows (scratch)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_vi
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof_vi -o vi -- python3 $ROOT/scratch/gpu_vi_small_latency.py > /dev/null 2>&1
python3 - <<'PY'
import glob, sqlite3
db = (glob.glob('/tmp/prof_vi/*.db') + glob.glob('/tmp/prof_vi/*/*.db'))[0]
c = sqlite3.connect(db)
tabs = [x[0] for x in c.execute("select name from sqlite_master where type='table'")]
sym = [t for t in tabs if 'info_kernel_symbol' in t][0]
disp = [t for t in tabs if 'kernel_dispatch' in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({disp})")]
q = f"select s.kernel_name, k.start, k.end, k.grid_size_x from {disp} k join {sym} s on k.kernel_id = s.id order by k.start"
rows = [(n, st, en, g) for n, st, en, g in c.execute(q)]
imu = [(en - st, g) for n, st, en, g in rows if 'k_imu' in n]
print("k_imu launches:", len(imu))
# last 16 launches = last 4 Solve(1) calls of the last problem
for d, g in imu[-16:]:
    print("grid %6d  %.1f us" % (g, d / 1e3))
PY
