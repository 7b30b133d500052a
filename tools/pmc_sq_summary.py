"""Per-kernel SQ activity from tools/pmc_sq.sh (rocprofv3 --pmc SQ_* pass): raw counter sums and ratios to SQ_WAVE_CYCLES.
  python tools/pmc_sq_summary.py gpurun_out/pmc_sq/sq_counter_collection.csv > profiles/r01_sq_activity.txt
Ratios are for comparing kernels with each other (the counters accumulate over shader engines and per-SIMD issue
slots; no absolute normalisation is attempted)."""
import collections
import csv
import re
import sys

d = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?_kernel)I", n)
    if m:
        n = m.group(1)
    n = n[:48]
    d[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        cnt[n] += 1
rows = sorted(((c.get("SQ_WAVE_CYCLES", 0), n, c) for n, c in d.items() if c.get("SQ_WAVE_CYCLES", 0) > 0), reverse=True)
print("# bench.py --steps 2 --warmup 1 --no-graph, one PMC pass; ratios relative to SQ_WAVE_CYCLES (x1e-2), MFMA busy relative to SQ_BUSY_CYCLES")
print("# flat = global_* / flat_* memory instructions (SQ_ACTIVE_INST_FLAT: on gfx950 the VMEM counter only sees buffer_* ops, which is why")
print("# round 2's vmem column read 0.00); parked = SQ_WAIT_ANY (s_waitcnt / barrier), stall = SQ_WAIT_INST_ANY (issue stalls)")
print("%-48s %6s %12s %8s %8s %8s %8s %8s %8s" % ("kernel", "calls", "wave_cycles", "valu", "lds", "flat", "parked", "stall", "mfma"))
for wc, n, c in rows[:40]:
    print("%-48s %6d %12.3e %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f" % (
        n, cnt[n], wc, 100 * c["SQ_ACTIVE_INST_VALU"] / wc, 100 * c["SQ_ACTIVE_INST_LDS"] / wc, 100 * c["SQ_ACTIVE_INST_FLAT"] / wc,
        100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc,
        100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c.get("SQ_BUSY_CYCLES", 1), 1)))
