"""Copy the summaries of gpurun_out/prof_round (tools/profile_round.sh) into profiles/ under a round tag and add up the
kernel trace per kernel family:  python tools/refresh_profiles.py r02_c"""
import collections
import csv
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_summary import base_name  # noqa: E402


def main():
    tag = sys.argv[1]
    src, dst = os.path.join(ROOT, "gpurun_out", "prof_round"), os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats_bench_b32.csv"))
    shutil.copy(os.path.join(src, "bench.jsonl"), os.path.join(dst, f"{tag}_bench_b32.jsonl"))
    shutil.copy(os.path.join(src, "per_shape.txt"), os.path.join(dst, f"{tag}_per_shape_bench_b32.txt"))
    shutil.copy(os.path.join(src, "pmc_traffic.json"), os.path.join(dst, tag.split("_")[0] + "_pmc_traffic.json"))
    families(tag, os.path.join(src, "kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_families_bench_b32.txt"),
             "PB_WGRAD_OVERLAP=0 (every kernel alone on the chip)")
    if os.path.exists(os.path.join(src, "kernel_stats_overlap.csv")):
        shutil.copy(os.path.join(src, "kernel_stats_overlap.csv"), os.path.join(dst, f"{tag}_kernel_stats_overlap_bench_b32.csv"))
        families(tag, os.path.join(src, "kernel_stats_overlap.csv"), os.path.join(dst, f"{tag}_kernel_families_overlap_bench_b32.txt"),
                 "default schedule (each block's weight gradients on a second HIP stream: kernels overlap, single durations stretch)")


def families(tag, stats_csv, out_path, how):
    rows = list(csv.DictReader(open(stats_csv)))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        b = base_name(r["Name"])
        agg[b][0] += int(r["Calls"])
        agg[b][1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    steps = 12   # bench.py --steps 5 --warmup 2: 2 warm-up + 5 timed + the 5 re-run with HIP events
    out = ["# rocprofv3 --kernel-trace --stats of `python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph`, " + how,
           "# (12 eager steps: 2 warm-up + 5 timed + the 5 re-run with HIP events); template instantiations of a kernel added up",
           f"# (tools/pmc_summary.base_name); full rows: {tag}_kernel_stats_bench_b32.csv",
           "%-34s %10s %12s %10s %7s" % ("kernel", "calls/step", "avg us", "ms/step", "share")]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append("%-34s %10.1f %12.2f %10.3f %6.1f%%" % (k, v[0] / steps, v[1] / v[0] / 1e3, v[1] / steps / 1e6, 100 * v[1] / tot))
    out.append("total kernel time per step: %.3f ms, %d launches per step" % (tot / steps / 1e6, sum(v[0] for v in agg.values()) / steps))
    open(out_path, "w").write("\n".join(out) + "\n")
    print("\n".join(out[3:9]))
    print(out[-1])


if __name__ == "__main__":
    main()
