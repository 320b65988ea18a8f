#!/bin/bash
# usage: tools/gaps.sh  -> per-frame timeline (start offset, duration, gap before) of the foveated frame's kernels
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
rm -rf /tmp/gp1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gp1 -o g -- python3 tools/stage_bench.py fov_pcheck_obb 12 > /dev/null 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("/tmp/gp1/g_kernel_trace.csv")))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")[:34]) for r in rows]
try:
    for r in csv.DictReader(open("/tmp/gp1/g_memory_copy_trace.csv")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy/" + r.get("Direction", "")))
except Exception as e:
    pass
ev.sort()
# last complete frame: from the last k_project (or tile_levels) backwards
idx = [i for i, e in enumerate(ev) if "k_tile_levels" in e[2]]
a, b = idx[-3], idx[-2]
t0 = ev[a][0]; prev_end = None
for s, e, n in ev[a:b]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    prev_end = max(prev_end or e, e)
print("frame span %.1f us" % ((ev[b][0] - t0) / 1e3))
PY
