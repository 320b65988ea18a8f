#!/bin/bash
# usage: tools/pmc_any.sh "<counters>" <script.py> [args] -> per-kernel average counter values for fr:: kernels
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
C="$1"; shift
rm -rf /tmp/pmc2
rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc2 -o p -- python3 "$@" > /dev/null 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("/tmp/pmc2/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "fr::" not in k: continue
    agg[k.replace("void ","")[:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
