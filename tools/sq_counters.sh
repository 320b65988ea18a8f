#!/bin/bash
# SQ counter passes over the foveated bench frames (run on the GPU box through gpurun):
#   tools/sq_counters.sh r02 [bench args]  ->  gpurun_out/profiles/r02_render_sq.json
# Each pass is its own rocprofv3 run (--kernel-trace --pmc only; 8 SQ slots + 2 GRBM slots per pass), the program
# directly after `--`. Per-launch averages for every fr:: kernel; derived figures for the blend kernels.
TAG=${1:-r02}; shift
ARGS=${@:---steps 12 --warmup 4 --no-extra --no-cpu-baseline}
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $OUT
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf /tmp/sq_$i
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/sq_$i -o p -- python3 bench.py $ARGS > /tmp/sq_$i.log 2>&1
done
SHA=$(sha256sum fov-3dgs_amd/libfovraster_hip.so | cut -c1-16)
python3 - "$TAG" "$ARGS" "$SHA" <<'PY'
import csv, collections, json, os, re, sys
tag, args, sha = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for i in (1, 2, 3):
    f = f"/tmp/sq_{i}/p_counter_collection.csv"
    if not os.path.exists(f):
        print("missing", f, open(f"/tmp/sq_{i}.log").read()[-2000:]); continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fr::" not in k: continue
        name = k.replace("void ", "").split("fr::")[1].split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kt = f"/tmp/sq_{i}/p_kernel_trace.csv"
    if i == 1 and os.path.exists(kt):
        for r in csv.DictReader(open(kt)):
            k = r["Kernel_Name"]
            if "fr::" not in k: continue
            name = k.replace("void ", "").split("fr::")[1].split("(")[0]
            dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for name, d in agg.items():
    # a counter row appears once per dispatch (summed over XCDs / SEs by rocprofv3) -- average per launch
    e = {c: round(sum(v) / len(v), 1) for c, v in d.items()}
    e["launches"] = max(len(v) for v in d.values())
    if dur.get(name): e["avg_duration_us_profiled"] = round(sum(dur[name]) / len(dur[name]) / 1e3, 2)
    wc, busy = e.get("SQ_WAVE_CYCLES"), e.get("SQ_BUSY_CYCLES")
    if wc:
        # SQ_WAVE_CYCLES / WAIT_* / ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md)
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM"):
            if c in e: e[c + "_per_WAVE_CYCLES"] = round(e[c] / wc, 4)
        if e.get("SQ_WAVES"): e["wave_cycles_x4_per_wave"] = round(4 * wc / e["SQ_WAVES"], 1)
    if e.get("GRBM_GUI_ACTIVE") and wc:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs (4.2 M for a 231 us kernel = 526 k cycles at 2.28 GHz each); the chip
        # has 1024 SIMDs, a wave64 VALU instruction occupies its SIMD for one quad-cycle
        cyc = e["GRBM_GUI_ACTIVE"] / 8.0
        e["gpu_cycles"] = round(cyc)
        if "avg_duration_us_profiled" in e: e["clock_GHz"] = round(cyc / e["avg_duration_us_profiled"] / 1e3, 3)
        e["occupancy_waves_per_simd"] = round(4 * wc / (cyc * 1024), 3)          # mean resident waves per SIMD (max 8)
        if "SQ_ACTIVE_INST_VALU" in e: e["valu_frac"] = round(4 * e["SQ_ACTIVE_INST_VALU"] / (cyc * 1024), 4)  # share of SIMD cycles with a VALU instruction executing
        if "SQ_INSTS_VALU" in e: e["valu_insts"] = round(e["SQ_INSTS_VALU"])
        if "SQ_WAIT_INST_ANY" in e: e["wait_inst_frac"] = round(e["SQ_WAIT_INST_ANY"] / wc, 4)   # wave-cycles waiting to issue
        if "SQ_WAIT_ANY" in e: e["wait_any_frac"] = round(e["SQ_WAIT_ANY"] / wc, 4)              # wave-cycles parked on s_waitcnt
        if "SQ_WAIT_INST_LDS" in e: e["lds_wait_frac"] = round(e["SQ_WAIT_INST_LDS"] / wc, 4)
    out[name] = e
doc = {"command": "rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py " + args + "  (three separate passes)",
       "lib_sha16": sha,
       "note": "per-launch averages; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; GRBM_GUI_ACTIVE = GPU cycles of the dispatch",
       "kernels": out}
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "profiles", f"{tag}_render_sq.json")
json.dump(doc, open(p, "w"), indent=1)
for n in out:
    if "render" in n: print(n, json.dumps(out[n]))
PY
