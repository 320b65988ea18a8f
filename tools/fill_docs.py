#!/usr/bin/env python3
"""(Re-)fill the bench numbers of DESIGN.md / README.md from profiles/r02_bench_line.json: python tools/fill_docs.py"""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_line.json")))
st, ex, rf = d["stages_ms"], d["extra"], d["roofline"]
pk = rf["per_kernel"]
vals = {
    "R2_VALUE_PACKED": f"{d['value_packed']:.0f}", "R2_VALUE": f"{d['value']:.0f}", "R2_MS": f"{d['ms_per_step']:.2f}",
    "R2_PROJECT": f"{st['project']:.3f}", "R2_BIN_GBS": f"{pk['bin']['alg_GBs'] / 1000:.2f}", "R2_BIN_FRAC": f"{pk['bin']['alg_GBs'] / 8000:.2f}",
    "R2_BIN": f"{st['bin']:.3f}", "R2_SCAN": f"{st['tile_scan']:.3f}", "R2_EMIT": f"{st['emit']:.3f}", "R2_SORT": f"{st['tile_sort']:.3f}",
    "R2_RENDER": f"{st['render']:.3f}", "R2_BLEND_GBS": f"{rf['blend']['achieved'] / 1000:.2f}", "R2_BLEND_FRAC": f"{rf['blend']['frac']:.2f}",
    "R2_NONFOV": f"{ex['nonfov_forward_fps']:.0f}", "R2_TFWD": f"{ex['train_fwd_ms']:.2f}", "R2_TLOSS": f"{ex['train_loss_fwd_ms']:.2f}",
    "R2_TBWD": f"{ex['train_bwd_ms']:.2f}", "R2_TSTEP": f"{ex['train_step_ms']:.2f}", "R2_CPU": f"{d['cpu_baseline']['value']:.2f}",
    "R2_RAWFWD": f"{ex['train_raw_fwd_ms']:.2f}", "R2_RAWLOSS": f"{ex['train_raw_loss_fwd_ms']:.2f}", "R2_RAWBWD": f"{ex['train_raw_bwd_ms']:.2f}",
    "R2_RAWSTEP": f"{ex['train_raw_step_ms']:.2f}",
}
# numbers sit between invisible markers: <!--R2_VALUE-->1238<!--/-->; a bare R2_VALUE (first fill) gets its markers here
for name in ("DESIGN.md", "README.md"):
    p = os.path.join(ROOT, name)
    s = open(p).read()
    for k in sorted(vals, key=len, reverse=True):
        s = re.sub(r"(?<![-A-Z_])" + k + r"(?![A-Z_-])", f"<!--{k}-->{vals[k]}<!--/-->", s)
        s = re.sub(r"<!--" + k + r"-->[^<]*<!--/-->", f"<!--{k}-->{vals[k]}<!--/-->", s)
    open(p, "w").write(s)
print(vals)
