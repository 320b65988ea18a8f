#!/bin/bash
# usage: tools/kstats_bench.sh [bench args]  -> per-kernel average durations (us) of the fr:: kernels over a bench.py run
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
rm -rf /tmp/ksb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksb -o k -- python3 bench.py "$@" > /tmp/ksb.log 2>&1
grep '^{"metric"' /tmp/ksb.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('stages_ms'), d.get('fwd_bwd_ms'), d.get('extra'))"
python3 - <<PY
import csv
rows = [r for r in csv.DictReader(open("/tmp/ksb/k_kernel_stats.csv")) if "fr::" in r["Name"]]
for r in sorted(rows, key=lambda r: r["Name"]):
    print("%-64s calls %4s avg %8.1f us  min %8.1f max %8.1f" % (r["Name"].replace("void ", "")[:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
