#!/bin/bash
# Regenerates the rocprofv3 summaries that profiles/ holds (run on the GPU box through gpurun):
#   tools/make_profiles.sh r05   ->  gpurun_out/profiles/r05_*
# One workload per file, so that a reader can recompute every roofline figure of the bench line from profiles/ alone:
# 1) r05_headline_kernel_stats.csv   kernel trace + stats of the HEADLINE frames only (bench.py --headline-only --no-extra: 9
#                                    warm-up + 5 x 63 timed foveated frames from the reference's tensors, two untimed stage passes)
#    r05_bench_kernel_stats.csv      ... of the default bench run (headline + packed + two-in-flight frames + extras)
#    r05_train_kernel_stats.csv      ... of the training step (bench.py --mode train)
# 2) r05_pmc.json                    two separate --pmc passes each (FETCH_SIZE, WRITE_SIZE) over the headline frames and over the
#                                    training step: bytes per launch and per frame AND the kernels' average durations (avg_ns: of
#                                    the un-counted stats pass 1; avg_ns_pmc: of the counter pass itself); calibration factors from
#                                    the k_pack_* launches (known byte counts) of the packed pass
# 3) r05_render_sq.json              SQ counter passes (tools/sq_counters.sh)
# The sha of the library build is recorded: bench.py quotes PMC bytes only for the build they were measured on.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $OUT
SHA=$(sha256sum fov-3dgs_amd/libfovraster_hip.so | cut -c1-16)
rm -rf /tmp/prof_h /tmp/prof_o /tmp/prof_a /tmp/prof_t /tmp/prof_f /tmp/prof_w /tmp/prof_pf /tmp/prof_pw /tmp/prof_tf /tmp/prof_tw
# (--serial-only: one frame on the GPU at a time -- a kernel's duration and counters are its own; the default run overlaps successive frames)
HEAD="--headline-only --no-extra --no-cpu-baseline --serial-only"
TRAIN="--mode train --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o h -- python3 bench.py $HEAD > /tmp/prof_h.log 2>&1
cp /tmp/prof_h/h_kernel_stats.csv $OUT/${TAG}_headline_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_o -o o -- python3 bench.py --headline-only --no-extra --no-cpu-baseline > /tmp/prof_o.log 2>&1
cp /tmp/prof_o/o_kernel_stats.csv $OUT/${TAG}_overlapped_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o b -- python3 bench.py --no-cpu-baseline > /tmp/prof_a.log 2>&1
cp /tmp/prof_a/b_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
grep '^{"metric"' /tmp/prof_a.log | tail -1 > $OUT/${TAG}_bench_line_profiled.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python3 bench.py $TRAIN > /tmp/prof_t.log 2>&1
cp /tmp/prof_t/t_kernel_stats.csv $OUT/${TAG}_train_kernel_stats.csv
grep '^{"metric"' /tmp/prof_t.log | tail -1 > $OUT/${TAG}_train_line_profiled.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o p -- python3 bench.py $HEAD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o p -- python3 bench.py $HEAD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_pf -o p -- python3 bench.py --packed-only --serial-only --no-extra --no-cpu-baseline --steps 18 --repeats 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_pw -o p -- python3 bench.py --packed-only --serial-only --no-extra --no-cpu-baseline --steps 18 --repeats 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_tf -o p -- python3 bench.py $TRAIN > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_tw -o p -- python3 bench.py $TRAIN > /dev/null 2>&1
python3 - "$TAG" "$SHA" <<'PY'
import csv, collections, json, os, re, sys
tag, sha = sys.argv[1], sys.argv[2]
P = 6_000_000


def kname(k):
    return k.replace("void ", "").split("fr::")[1].split("(")[0] if "fr::" in k else None


def base_of(full):
    return re.sub(r"<.*", "", full)


def stats_ns(path):
    """kernel name -> (calls, average ns) of a --stats pass"""
    out = {}
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            n = kname(r["Name"])
            if n:
                out[n] = (int(r["Calls"]), float(r["AverageNs"]))
    return out


def pmc_pass(dirs, frames_kernel, group_by_base):
    """-> {kernel: {FETCH_SIZE_KiB (per launch), ..._per_frame, launches_per_frame, avg_ns_pmc}}. A frame = one launch of
    `frames_kernel`. group_by_base: template instances of one kernel are added up (the headline pass runs one instance each)."""
    out = collections.defaultdict(dict)
    for d, c in dirs:
        agg, dur = collections.defaultdict(list), collections.defaultdict(list)
        f = d + "/p_counter_collection.csv"
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            n = kname(r["Kernel_Name"])
            if not n or r["Counter_Name"] != c:
                continue
            agg[base_of(n) if group_by_base else n].append(float(r["Counter_Value"]))
        kt = d + "/p_kernel_trace.csv"
        if os.path.exists(kt):
            for r in csv.DictReader(open(kt)):
                n = kname(r["Kernel_Name"])
                if n:
                    dur[base_of(n) if group_by_base else n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        nframes = max(1, len(agg.get(frames_kernel, [])) or max((len(v) for k, v in agg.items() if k.startswith(frames_kernel)), default=1))
        for name, v in agg.items():
            out[name][c + "_KiB"] = round(sum(v) / len(v))
            out[name][c + "_KiB_per_frame"] = round(sum(v) / nframes)
            out[name]["launches_per_frame"] = round(len(v) / nframes, 2)
            if dur.get(name):
                out[name]["avg_ns_pmc"] = round(sum(dur[name]) / len(dur[name]))
    return out


head = pmc_pass((("/tmp/prof_f", "FETCH_SIZE"), ("/tmp/prof_w", "WRITE_SIZE")), "k_emit", True)
packed = pmc_pass((("/tmp/prof_pf", "FETCH_SIZE"), ("/tmp/prof_pw", "WRITE_SIZE")), "k_emit", True)
train = pmc_pass((("/tmp/prof_tf", "FETCH_SIZE"), ("/tmp/prof_tw", "WRITE_SIZE")), "k_render_bwd", False)
# un-counted durations of the same workloads (pass 1)
hs, ts = stats_ns("/tmp/prof_h/h_kernel_stats.csv"), stats_ns("/tmp/prof_t/t_kernel_stats.csv")
for name, e in head.items():
    tot = [(c, ns) for n, (c, ns) in hs.items() if base_of(n) == name]
    if tot:
        e["avg_ns"] = round(sum(c * ns for c, ns in tot) / sum(c for c, _ in tot))
for name, e in train.items():
    if name in ts:
        e["avg_ns"] = round(ts[name][1])
kernels = dict(head)
for name, e in packed.items():
    if name in ("k_project", "k_bin") or name.startswith("k_pack"):
        kernels[name + ("_packed" if not name.startswith("k_pack") else "")] = e
# calibration on launches whose byte counts are known (bytes per Gaussian: fovraster.h packed_* rows and their inputs)
known = {"k_pack_geom": (60, 64), "k_pack_cull": (40, 16), "k_pack_colour": (228, 256)}
cal = {}
for name, (rd, wr) in known.items():
    e = kernels.get(name)
    if e and "FETCH_SIZE_KiB" in e and "WRITE_SIZE_KiB" in e:
        cal[name] = {"read_bytes": rd * P, "FETCH_SIZE_bytes": e["FETCH_SIZE_KiB"] * 1024, "read_factor": round(rd * P / (e["FETCH_SIZE_KiB"] * 1024), 3),
                     "written_bytes": wr * P, "WRITE_SIZE_bytes": e["WRITE_SIZE_KiB"] * 1024, "write_factor": round(wr * P / (e["WRITE_SIZE_KiB"] * 1024), 3)}
wide = [cal[n] for n in ("k_pack_geom", "k_pack_cull") if n in cal]
calibration = {"launches": cal,
               "fetch_factor": round(sum(c["read_bytes"] for c in wide) / sum(c["FETCH_SIZE_bytes"] for c in wide), 3) if wide else 2.0,
               "write_factor": round(sum(c["written_bytes"] for c in cal.values()) / sum(c["WRITE_SIZE_bytes"] for c in cal.values()), 3) if cal else 1.0,
               "note": "fetch_factor from the 16-byte-per-lane streaming reads of k_pack_geom / k_pack_cull (gfx950 FETCH_SIZE counts those at half size, MI355X_MICROARCH.md); k_pack_colour's mix of 4- and 16-byte reads shows a smaller factor: for gather-heavy kernels 2 x FETCH is an upper bound"}
doc = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and a separate WRITE_SIZE pass) -- python3 bench.py --headline-only --no-extra --no-cpu-baseline  |  --packed-only ... --steps 18 --repeats 1  |  --mode train --steps 20 --warmup 5",
       "lib_sha16": sha, "layout": "kernels = the headline frames (the reference's tensors; template instances of one kernel added up); *_packed = the static-model instances; train = the training step's kernels by full name",
       "note": "KiB as rocprofv3 reports them: per-launch averages and per-frame totals (a frame = one k_emit launch; a training step = one k_render_bwd launch); avg_ns = average kernel duration of the same workload WITHOUT counters (r05_headline_kernel_stats.csv / r05_train_kernel_stats.csv), avg_ns_pmc = inside the counter pass",
       "calibration": calibration, "kernels": kernels, "train": train}
json.dump(doc, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "profiles", f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps(calibration)[:400])
for n in ("k_project", "k_bin", "k_emit", "k_render_fov"):
    print(n, kernels.get(n))
PY
tools/sq_counters.sh $TAG --steps 27 --warmup 9 --no-extra --no-cpu-baseline --headline-only --serial-only --repeats 1 | tail -3
# 4) the same on S-6M-T (the list-consuming cloud): kernel stats of its foveated frames and of its training step, the whole bench line
#    on it (counts: blend pairs of every blend kernel), SQ passes over both -> ${TAG}_valu_per_pair.json: vector instructions per
#    blended (band, entry) pair of k_render_fov, k_render<1,2> and k_render_bwd on both clouds
rm -rf /tmp/prof_th /tmp/prof_tt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_th -o h -- python3 bench.py --cloud S-6M-T $HEAD > /tmp/prof_th.log 2>&1
cp /tmp/prof_th/h_kernel_stats.csv $OUT/${TAG}_translucent_headline_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tt -o t -- python3 bench.py --cloud S-6M-T $TRAIN > /tmp/prof_tt.log 2>&1
cp /tmp/prof_tt/t_kernel_stats.csv $OUT/${TAG}_translucent_train_kernel_stats.csv
python3 bench.py --cloud S-6M-T --no-cpu-baseline 2> /dev/null | grep '^{"metric"' | tail -1 > $OUT/${TAG}_bench_line_translucent.json
tools/sq_counters.sh ${TAG}T --cloud S-6M-T --steps 27 --warmup 9 --no-extra --no-cpu-baseline --headline-only --serial-only --repeats 1 | tail -1
tools/sq_counters.sh ${TAG}train --mode train --steps 12 --warmup 4 | tail -1
tools/sq_counters.sh ${TAG}Ttrain --cloud S-6M-T --mode train --steps 12 --warmup 4 | tail -1
python3 - "$TAG" <<'PY'
import json, os, sys
tag = sys.argv[1]
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "profiles")
ld = lambda n: json.load(open(os.path.join(out, n)))
line, line_t = ld(f"{tag}_bench_line_profiled.json"), ld(f"{tag}_bench_line_translucent.json")
sq = {k: ld(f"{tag}{k}_render_sq.json")["kernels"] for k in ("", "T", "train", "Ttrain")}
def find(d, prefix):
    return next((v for n, v in d.items() if n.startswith(prefix)), {})
rows = {}
for cloud, l, f, t in (("S-6M", line, sq[""], sq["train"]), ("S-6M-T", line_t, sq["T"], sq["Ttrain"])):
    c = l["roofline"].get("train", {}).get("counts", {})
    for kern, pairs, e in (("k_render_fov", l["config"].get("blend_pairs"), find(f, "k_render_fov")),
                           ("k_render<1,2>", c.get("blend_pairs_fwd"), find(t, "k_render<1")),
                           ("k_render_bwd", c.get("blend_pairs_bwd"), find(t, "k_render_bwd"))):
        if pairs and e.get("SQ_INSTS_VALU"):
            rows[f"{cloud} {kern}"] = dict(blend_pairs=pairs, valu_insts=round(e["SQ_INSTS_VALU"]), valu_insts_per_pair=round(e["SQ_INSTS_VALU"] / pairs, 1),
                                           salu_insts_per_pair=round(e.get("SQ_INSTS_SALU", 0) / pairs, 1), valu_frac=e.get("valu_frac"),
                                           occupancy_waves_per_simd=e.get("occupancy_waves_per_simd"), avg_duration_us_profiled=e.get("avg_duration_us_profiled"))
json.dump(dict(note="vector instructions per blended (band of eight rows, list entry) pair: SQ_INSTS_VALU of the kernel's launch (tools/sq_counters.sh, "
                    "three separate --pmc passes) / the pairs its waves evaluated (fr_forward_args.blend_pairs / fr_backward_args.blend_pairs, from the "
                    "bench line of the same build); a pair = 128 pixel updates", kernels=rows), open(os.path.join(out, f"{tag}_valu_per_pair.json"), "w"), indent=1)
print(json.dumps(rows)[:1500])
PY
head -25 $OUT/${TAG}_headline_kernel_stats.csv | cut -c1-150
cut -c1-600 $OUT/${TAG}_bench_line_profiled.json
