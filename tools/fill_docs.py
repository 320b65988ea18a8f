#!/usr/bin/env python3
"""(Re-)fill the bench numbers of DESIGN.md / README.md from profiles/r03_bench_line.json: python tools/fill_docs.py"""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_line.json")))
st, ex, rf = d["stages_ms"], d["extra"], d["roofline"]
pk = rf["per_kernel"]
vals = {
    "R3_VALUE_PACKED": f"{d['value_packed']:.0f}", "R3_VALUE": f"{d['value']:.0f}", "R3_MS": f"{d['ms_per_step']:.3f}",
    "R3_SPREAD": f"{d['value_spread'][0]:.0f}–{d['value_spread'][1]:.0f}",
    "R3_LEVELS": f"{st['tile_levels']:.3f}", "R3_PROJECT": f"{st['project']:.3f}", "R3_BIN_GBS": f"{rf['achieved'] / 1000:.2f}",
    "R3_BIN_FRACT": f"{rf.get('frac_traffic', 0):.2f}", "R3_BIN_FRAC": f"{rf['frac']:.2f}", "R3_BIN_TRAFFIC": f"{(rf.get('traffic') or 0) / 1e9:.2f}",
    "R3_BIN": f"{st['bin']:.3f}", "R3_SCAN": f"{st['tile_scan']:.3f}", "R3_EMIT": f"{st['emit']:.3f}", "R3_SORT": f"{st['tile_sort']:.3f}",
    "R3_RENDER": f"{st['render']:.3f}", "R3_BLEND_FRACT": f"{rf['blend'].get('frac_traffic', 0):.2f}", "R3_BLEND_FRAC": f"{rf['blend']['frac']:.2f}",
    "R3_NONFOV": f"{ex['nonfov_forward_fps']:.0f}", "R3_TSTEP": f"{ex['train_step_ms']:.2f}", "R3_CPU": f"{d['cpu_baseline']['value']:.2f}",
    "R3_RAWSTEP": f"{ex['train_raw_step_ms']:.2f}", "R3_REFSHAPED": f"{ex['reference_shaped_fps']:.0f}",
}
assert rf["kernel"] == "bin", "DESIGN.md section 5 names the binning stage as the slowest one"
# numbers sit between invisible markers: <!--R3_VALUE-->1238<!--/-->; a bare R3_VALUE (first fill) gets its markers here
for name in ("DESIGN.md", "README.md"):
    p = os.path.join(ROOT, name)
    s = open(p).read()
    for k in sorted(vals, key=len, reverse=True):
        s = re.sub(r"(?<![-A-Z_])" + k + r"(?![A-Z_-])", f"<!--{k}-->{vals[k]}<!--/-->", s)
        s = re.sub(r"<!--" + k + r"-->[^<]*<!--/-->", f"<!--{k}-->{vals[k]}<!--/-->", s)
    open(p, "w").write(s)
print(vals)
