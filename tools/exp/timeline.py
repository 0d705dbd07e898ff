#!/usr/bin/env python3
"""Timeline of ONE steady-state batched step from a rocprofv3 rocpd database: tools/exp/timeline.py results.db [step_index]
Prints every kernel launch between two consecutive k_quadtree starts: name, start offset (us), duration (us)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
q = [s for n, s, e in rows if "k_resize_regions" in n] or [s for n, s, e in rows if "k_load_level0" in n]
i = int(sys.argv[2]) if len(sys.argv) > 2 else len(q) // 2
t0, t1 = q[i], q[i + 1]
print(f"step {i}: {1e-3 * (t1 - t0):.1f} us between the starts of two steps")
for n, s, e in rows:
    if s >= t0 - 1500000 and s < t1:
        m = re.search(r"(k_\w+)", n)
        print(f"{(m.group(1) if m else n[:30]):20s} start {1e-3 * (s - t0):9.1f}  dur {1e-3 * (e - s):8.1f}  end {1e-3 * (e - t0):9.1f}")
