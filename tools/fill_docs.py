#!/usr/bin/env python3
"""Fill the R2_* placeholders of DESIGN.md / README.md from profiles/r02_bench_line.json (python tools/fill_docs.py), or
re-fill an already filled copy from the git version that still has them (python tools/fill_docs.py --from-git REV)."""
import json, os, re, subprocess, sys
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
}
rev = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "--from-git" else None
for name in ("DESIGN.md", "README.md"):
    p = os.path.join(ROOT, name)
    s = subprocess.check_output(["git", "show", f"{rev}:{name}"], cwd=ROOT, text=True) if rev else open(p).read()
    for k in sorted(vals, key=len, reverse=True):
        s = s.replace(k, vals[k])
    left = re.findall(r"R2_[A-Z_]+", s)
    assert not left, left
    open(p, "w").write(s)
print(vals)
