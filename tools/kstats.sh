#!/bin/bash
# usage: tools/kstats.sh [stage_bench args]  -> per-kernel average durations (us) of the fr:: kernels
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
rm -rf /tmp/ks1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -o k -- python3 tools/frames.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("/tmp/ks1/k_kernel_stats.csv")):
    if "fr::" in r["Name"]:
        print("%-40s calls %4s avg %8.1f us  min %8.1f" % (r["Name"].replace("void ", "")[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
