#!/bin/bash
# Regenerates the rocprofv3 summaries that profiles/ holds (run on the GPU box through gpurun):
#   tools/make_profiles.sh r02   ->  gpurun_out/profiles/r02_bench_kernel_stats.csv, r02_bench_line.json, r02_pmc.json, r02_render_sq.json
# 1) kernel trace + stats of the default bench run (headline frames, packed frames, extras; no CPU baseline)
# 2) two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over the SAME foveated frames (same --steps / --warmup, fixed gaze
#    set), per-launch averages and per-frame totals, packed and unpacked template instances kept apart; calibration
#    factors from the k_pack_* launches (known byte counts) of the same pass
# 3) SQ counter passes (tools/sq_counters.sh)
# The sha of the library build is recorded: bench.py quotes PMC bytes only for the build they were measured on.
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $OUT
SHA=$(sha256sum fov-3dgs_amd/libfovraster_hip.so | cut -c1-16)
rm -rf /tmp/prof_a /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o b -- python3 bench.py --no-cpu-baseline > /tmp/prof_a.log 2>&1
cp /tmp/prof_a/b_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
cp /tmp/prof_a/b_domain_stats.csv $OUT/${TAG}_bench_domain_stats.csv 2>/dev/null
grep '^{"metric"' /tmp/prof_a.log | tail -1 > $OUT/${TAG}_bench_line_profiled.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o p -- python3 bench.py --no-cpu-baseline --no-extra > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o p -- python3 bench.py --no-cpu-baseline --no-extra > /dev/null 2>&1
python3 - "$TAG" "$SHA" <<'PY'
import csv, collections, json, os, re, sys
tag, sha = sys.argv[1], sys.argv[2]
P = 6_000_000
out = collections.defaultdict(dict)
raw = {}
for d, c in (("/tmp/prof_f", "FETCH_SIZE"), ("/tmp/prof_w", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d + "/p_counter_collection.csv")):
        k = r["Kernel_Name"]
        if "fr::" not in k or r["Counter_Name"] != c: continue
        full = k.replace("void ", "").split("fr::")[1].split("(")[0]
        base = re.sub(r"<.*", "", full)
        # template instances of the packed static-model layout: k_bin<V, LDSH, PACKED[, CROW]>, k_project<V, PACKED>
        targs = [t.strip() for t in full[full.index("<") + 1:full.rindex(">")].split(",")] if "<" in full else []
        packed = (base == "k_bin" and len(targs) >= 3 and targs[2] == "true") or (base == "k_project" and len(targs) >= 2 and targs[1] == "true")
        agg[base + ("_packed" if packed else "")].append(float(r["Counter_Value"]))
    raw[c] = agg
    for name, v in agg.items():
        # launches of one frame: the per-tile sort runs several launches per frame; a frame = one k_tile_scan launch of
        # the same layout (packed and unpacked instances run in different frames)
        frames = len(v) if not name.startswith(("k_tile_msort", "k_split_long")) else max(1, len(agg.get("k_tile_scan", [1])))
        out[name][c + "_KiB"] = round(sum(v) / len(v))
        out[name][c + "_KiB_per_frame"] = round(sum(v) / frames)
        out[name]["launches_per_frame"] = round(len(v) / frames, 2)
# calibration on launches whose byte counts are known (bytes per Gaussian: fovraster.h packed_* rows and their inputs)
known = {"k_pack_geom": (60, 64), "k_pack_cull": (40, 16), "k_pack_colour": (228, 256)}
cal = {}
for name, (rd, wr) in known.items():
    if name in out and "FETCH_SIZE_KiB" in out[name]:
        cal[name] = {"read_bytes": rd * P, "FETCH_SIZE_bytes": out[name]["FETCH_SIZE_KiB"] * 1024, "read_factor": round(rd * P / (out[name]["FETCH_SIZE_KiB"] * 1024), 3),
                     "written_bytes": wr * P, "WRITE_SIZE_bytes": out[name]["WRITE_SIZE_KiB"] * 1024, "write_factor": round(wr * P / (out[name]["WRITE_SIZE_KiB"] * 1024), 3)}
wide = [cal[n] for n in ("k_pack_geom", "k_pack_cull") if n in cal]
calibration = {"launches": cal,
               "fetch_factor": round(sum(c["read_bytes"] for c in wide) / sum(c["FETCH_SIZE_bytes"] for c in wide), 3) if wide else 2.0,
               "write_factor": round(sum(c["written_bytes"] for c in cal.values()) / sum(c["WRITE_SIZE_bytes"] for c in cal.values()), 3) if cal else 1.0,
               "note": "fetch_factor from the 16-byte-per-lane streaming reads of k_pack_geom / k_pack_cull (gfx950 FETCH_SIZE counts those at half size, MI355X_MICROARCH.md); k_pack_colour's mix of 4- and 16-byte reads shows a smaller factor: for gather-heavy kernels 2 x FETCH is an upper bound"}
doc = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and a separate WRITE_SIZE pass) -- python3 bench.py --no-cpu-baseline --no-extra",
       "lib_sha16": sha, "layout": "headline frames = the reference's tensors (k_bin<3,true,false,true>, k_project<3,false>); *_packed = the static-model instances",
       "note": "per-launch averages (KiB as rocprofv3 reports them) and per-frame totals over the 9 + 63 frames of each layout",
       "calibration": calibration, "kernels": out}
json.dump(doc, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "profiles", f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps(calibration)[:600])
PY
tools/sq_counters.sh $TAG --steps 27 --warmup 9 --no-extra --no-cpu-baseline --packed-only | tail -3
head -30 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-150
cat $OUT/${TAG}_bench_line_profiled.json | cut -c1-400
