#!/bin/bash
# Regenerates the rocprofv3 summaries that profiles/ holds (run on the GPU box through gpurun):
#   tools/make_profiles.sh r01   ->  gpurun_out/profiles/r01_bench_kernel_stats.csv, r01_bench_domain_stats.csv, r01_pmc.json
# 1) kernel trace + stats of the default bench run (foveated frames + extras, no CPU baseline)
# 2) two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short foveated-only run, per-launch averages
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $OUT
rm -rf /tmp/prof_a /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o b -- python3 bench.py --no-cpu-baseline > /tmp/prof_a.log 2>&1
grep "^{\"metric\"" /tmp/prof_a.log | tail -1 > $OUT/${TAG}_bench_line.json
cp /tmp/prof_a/b_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
cp /tmp/prof_a/b_domain_stats.csv $OUT/${TAG}_bench_domain_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
python3 - <<PY
import csv, collections, json, re
out = collections.defaultdict(dict)
for d, c in (("/tmp/prof_f", "FETCH_SIZE"), ("/tmp/prof_w", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d + "/p_counter_collection.csv")):
        k = r["Kernel_Name"]
        if "fr::" not in k or r["Counter_Name"] != c: continue
        name = re.sub(r"<.*", "", k.replace("void ", "").split("fr::")[1].split("(")[0])
        agg[name].append(float(r["Counter_Value"]))
    frames = max(1, len(agg.get("k_project", [1])))
    for name, v in agg.items():
        out[name][c + "_KiB"] = round(sum(v) / len(v))            # per launch
        out[name][c + "_KiB_per_frame"] = round(sum(v) / frames)  # all launches of one frame (the sort runs one launch per size class)
        out[name]["launches_per_frame"] = round(len(v) / frames, 2)
doc = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and a separate WRITE_SIZE pass) -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra",
       "note": "per-launch averages over the foveated frames; KiB as reported by rocprofv3 (gfx950: FETCH_SIZE counts wide 16 B/lane reads at half size; WRITE_SIZE uncalibrated)",
       "kernels": out}
json.dump(doc, open("$OUT/${TAG}_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
head -14 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-150
cat $OUT/${TAG}_bench_line.json | cut -c1-600
