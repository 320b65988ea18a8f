"""Collects the measured differences of every image / gradient comparison the parity tests make and writes them at the end
of the session, so that the numbers behind "within tolerance" are kept, not just pass/fail. Two files, so that a CPU
session can never overwrite what a GPU session measured (round 3 lost its whole-frame report that way):
  tests/parity_report_gpu.json   written only when a test marked `gpu` recorded a comparison (HIP path vs oracle)
  tests/parity_report_cpu.json   everything a session without GPU comparisons recorded (oracle vs golden vectors)
(+ the same file under gpurun_out/ when that directory exists, which is how a GPU box's report comes home)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_entries = []
_gpu_session = False  # a test marked `gpu` has recorded something (tests/conftest.py tells us which test is running)
current_test_is_gpu = False


def record(kind, name, **values):
    global _gpu_session
    _gpu_session = _gpu_session or current_test_is_gpu
    clean = {k: (float(v) if isinstance(v, (np.floating, float)) else int(v) if isinstance(v, (np.integer, int)) else v)
             for k, v in values.items()}
    _entries.append(dict(kind=kind, name=name, **clean))


def image_stats(got, want):
    d = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
    return dict(max_abs=float(d.max()) if d.size else 0.0, frac_gt_1e4=float(np.mean(d > 1e-4)) if d.size else 0.0,
                frac_gt_1e5=float(np.mean(d > 1e-5)) if d.size else 0.0, n=int(d.size))


def flush():
    if not _entries:
        return
    worst_img = max((e["max_abs"] for e in _entries if e["kind"] == "image"), default=None)
    worst_frac = max((e["frac_gt_1e4"] for e in _entries if e["kind"] == "image"), default=None)
    worst_grad = max((e["frac_bad"] for e in _entries if e["kind"] == "grad"), default=None)
    doc = dict(summary=dict(comparisons=len(_entries), worst_image_max_abs=worst_img, worst_image_frac_gt_1e4=worst_frac,
                            worst_grad_frac_outside_tolerance=worst_grad),
               entries=_entries)
    doc["summary"]["session"] = "gpu" if _gpu_session else "cpu"
    fname = "parity_report_gpu.json" if _gpu_session else "parity_report_cpu.json"
    for d in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, fname), "w") as f:
                    json.dump(doc, f, indent=1)
            except OSError:
                pass
