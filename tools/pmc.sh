#!/bin/bash
# usage: tools_pmc.sh "<counters>"  -> per-kernel average counter values for fr:: kernels
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
rm -rf /tmp/pmc1
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d /tmp/pmc1 -o p1 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("/tmp/pmc1/p1_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "fr::" not in k: continue
    agg[k.replace("void ","")[:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
