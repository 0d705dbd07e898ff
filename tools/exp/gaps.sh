# on the GPU box: kernel-trace of the plain step loop only (no host_io / cpu legs), then the busy-vs-span summary and the per-step gap list
export TMPDIR=/tmp
rm -rf gpurun_out/pg
rocprofv3 --kernel-trace -d gpurun_out/pg -- python3 tools/step_time.py 512 40 > gpurun_out/gaps_step.txt 2>&1
DB=$(find gpurun_out/pg -name "*.db" | head -1)
python3 tools/kernel_stats_from_db.py $DB | tail -1
python3 - $DB <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# the last 10 steps: find starts of k_load_level0
starts = [i for i, r in enumerate(rows) if "k_load_level0" in r[0]]
a, b = starts[-11], starts[-1]
seg = rows[a:b]
t0 = seg[0][1]
span = seg[-1][2] - t0
print("10 steps span ms", span / 1e6)
# timeline of one step
s1 = rows[starts[-3]:starts[-2]]
base = s1[0][1]
for n, s, e in s1:
    print(f"{n.split('(')[0].replace('void ','').replace('orbfe::',''):22s} {(s-base)/1e6:8.3f} -> {(e-base)/1e6:8.3f}  ({(e-s)/1e6:.3f})")
PY
rm -rf gpurun_out/pg
