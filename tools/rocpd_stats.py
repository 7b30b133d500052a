"""Per-kernel statistics from a rocprofv3 rocpd database (ROCm 7.2 writes *_results.db by default):
  python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--steps N] [--csv out.csv] [--by-grid]
Prints calls, total and average duration per kernel name (template arguments kept, parameter lists dropped)."""
import argparse
import collections
import re
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=1, help="divide calls/time by this many steps")
    ap.add_argument("--csv", default=None)
    ap.add_argument("--by-grid", action="store_true", help="separate rows per grid size (one row per layer shape)")
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    c = sqlite3.connect(args.db)
    rows = c.execute("select name, grid_x, start, end from kernels").fetchall()
    agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for name, gx, s, e in rows:
        name = re.sub(r"^void ", "", name)
        name = name.replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*$", "", name)
        key = (name, gx) if args.by_grid else (name, 0)
        d = (e - s) / 1e3
        a = agg[key]
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"{len(rows)} dispatches, {tot / 1e3 / args.steps:.3f} ms of kernel time per step ({args.steps} steps)")
    out = sorted(agg.items(), key=lambda kv: -kv[1][1])
    for (name, gx), a in out[:args.top]:
        g = f" grid {gx}" if args.by_grid else ""
        print("%-86s%s calls/step %6.1f  %9.1f us/step  avg %8.1f us  %5.1f%%" % (
            name[:86], g, a[0] / args.steps, a[1] / args.steps, a[1] / a[0], 100 * a[1] / tot))
    if args.csv:
        with open(args.csv, "w") as f:
            f.write("Name,Grid,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
            for (name, gx), a in out:
                f.write('"%s",%d,%d,%d,%.1f,%d,%d,%.3f\n' % (name, gx, a[0], a[1] * 1e3, a[1] * 1e3 / a[0], a[2] * 1e3, a[3] * 1e3, 100 * a[1] / tot))


if __name__ == "__main__":
    main()
