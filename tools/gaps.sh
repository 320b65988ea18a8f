#!/bin/bash
# usage: tools/gaps.sh [frames.py args]  -> timeline (start offset, duration, gap before) of one frame's kernels, and the
# mean busy / gap time per frame over all profiled frames
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
rm -rf /tmp/gp1
rocprofv3 --kernel-trace --output-format csv -d /tmp/gp1 -o g -- python3 tools/frames.py "$@" > /tmp/gp1.log 2>&1
tail -1 /tmp/gp1.log
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("/tmp/gp1/g_kernel_trace.csv")))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("fr::", "")[:40]) for r in rows)
# frames start at the first kernel after a k_render*: take the fills at the head with them
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_project")]
def frame_bounds(k):
    a = starts[k]
    while a > 0 and not ev[a - 1][2].startswith("k_render"): a -= 1
    return a
fb = [frame_bounds(k) for k in range(len(starts))]
a, b = fb[-3], fb[-2]
t0 = ev[a][0]; prev_end = None
for s, e, n in ev[a:b]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    prev_end = max(prev_end or e, e)
print("frame span %.1f us" % ((ev[b][0] - t0) / 1e3))
# all steady-state frames: union of busy intervals vs span
spans, busy = [], []
for k in range(10, len(fb) - 1):
    a, b = fb[k], fb[k + 1]
    iv = sorted((s, e) for s, e, n in ev[a:b])
    tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    tot += ce - cs
    spans.append((ev[b][0] - ev[a][0]) / 1e3); busy.append(tot / 1e3)
import statistics as st
print("frames %d: span mean %.1f us, GPU busy (union of kernels) %.1f us, idle %.1f us" % (len(spans), st.mean(spans), st.mean(busy), st.mean(spans) - st.mean(busy)))
PY
