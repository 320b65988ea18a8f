#!/bin/bash
# usage: tools/kstats_any.sh <script.py> [args]  -> per-kernel totals of one profiled run (top 25 by time)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
rm -rf /tmp/ks2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2 -o k -- python3 "$@" > /tmp/ks2.log 2>&1
tail -2 /tmp/ks2.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("/tmp/ks2/k_kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:25]:
    print("%-60s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"].replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
