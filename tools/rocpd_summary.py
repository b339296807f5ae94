#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (…_results.db) as text: the per-kernel --stats table plus,
for one kernel, the mean duration of its last K dispatches (= the timed region of bench.py).

usage: python tools/rocpd_summary.py <results.db> [kernel-substring] [last_k]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "tick_kernel"
    last_k = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    c = sqlite3.connect(db)
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}")
    print(f"{'kernel':<70} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}")
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{name[:70]:<70} {calls:>6} {total:>12.1f} {avg:>10.2f} {pct:>6.2f}")
    rows = c.execute("select duration, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x "
                     "from kernels where name like ? order by start", (f"%{pat}%",)).fetchall()
    if rows:
        d = [r[0] / 1e3 for r in rows]
        k = last_k if 0 < last_k <= len(d) else len(d)
        tail = d[-k:]
        print(f"\n# {pat}: {len(d)} dispatches; last {k}: mean {sum(tail)/k:.2f} us, min {min(tail):.2f}, max {max(tail):.2f}")
        r = rows[-1]
        print(f"# resources: vgpr {r[1]} agpr {r[2]} sgpr {r[3]} lds {r[4]} grid {r[5]} block {r[6]}")


if __name__ == "__main__":
    main()
