#!/bin/bash
# usage: tools/pmc_stage.sh "<counters>" [stage_bench args]  -> per-kernel average counter values (fr:: kernels)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
rm -rf /tmp/pmc1
C="$1"; shift
rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc1 -o p1 -- python3 tools/stage_bench.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("/tmp/pmc1/p1_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "fr::" not in k: continue
    agg[k.replace("void ","")[:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
