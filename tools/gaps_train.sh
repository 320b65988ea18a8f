#!/bin/bash
# usage: tools/gaps_train.sh [bench args] -> timeline of one training step (bench.py --mode train): start offset, duration, idle gap
# before every kernel / copy of the last complete step
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
rm -rf /tmp/gp2
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gp2 -o g -- python3 bench.py --mode train --steps 12 --warmup 3 "$@" > /dev/null 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("/tmp/gp2/g_kernel_trace.csv")))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")[:60]) for r in rows]
try:
    for r in csv.DictReader(open("/tmp/gp2/g_memory_copy_trace.csv")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy/" + r.get("Direction", "")))
except Exception as e:
    pass
ev.sort()
idx = [i for i, e in enumerate(ev) if "k_project" in e[2]]
a, b = idx[-3], idx[-2]
# a step starts a little before k_project (activations / fills): walk back to the previous step's last kernel
while a > 0 and "k_preprocess_bwd" not in ev[a - 1][2] and "k_activate_bwd" not in ev[a - 1][2]: a -= 1
while b > 0 and "k_preprocess_bwd" not in ev[b - 1][2] and "k_activate_bwd" not in ev[b - 1][2]: b -= 1
t0 = ev[a][0]; prev_end = None; busy = 0; idle = 0
for s, e, n in ev[a:b]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    if gap > 0: idle += gap
    print("%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    prev_end = max(prev_end or e, e)
print("step span %.1f us, idle gaps %.1f us" % ((ev[b][0] - t0) / 1e3, idle))
PY
