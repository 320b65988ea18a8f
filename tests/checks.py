"""Tolerance checks of the GPU parity tests. Every call records what it measured (tests/parity_report.py).

Tolerances (fp32 path; BASELINE north_star: per-pixel match within 1e-4):
  * image: |diff| <= 1e-4 on all but `frac` (1e-6) of the pixels and <= `hard` (2e-3) everywhere. The blend thresholds
    (alpha < 1/255, T < 1e-4, power < -4.5, power > 0) are discontinuous, and the HIP kernel uses the hardware exp2
    and fused multiply-adds while the oracle follows the reference's literal fp32 expression, so a pair that sits
    within an ulp of a threshold may flip; each flip moves a pixel by at most ~alpha (<= 1.2e-2 at the -4.5 cutoff).
    The measured fraction and maximum of every comparison are kept in tests/parity_report.json.
  * gradients: float atomics have no defined order in the reference either; the oracle sums in double.
    |diff| <= rtol * max(1, |ref|) + 1e-5 * max|ref| on all but `outlier_frac` of the entries (threshold flips).
"""
import os

import numpy as np

from tests import parity_report

# Bounds a few times above the worst case measured over the whole GPU suite (tests/parity_report.json, 132 comparisons):
# every image is within 1e-4 at EVERY pixel except one comparison (4096x2160, 26.5 M values) where 2.3e-7 of them -- a
# threshold flip -- are up to 7.1e-4 off; gradient entries outside rtol: at most 4.7e-5 of a tensor.
IMAGE_FRAC = 1e-6   # share of pixels allowed above 1e-4
IMAGE_HARD = 2e-3   # ... and how far those may be off
GRAD_OUTLIERS = 2e-4


def _where():
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::")[-1]


def check_image(got, want, frac=IMAGE_FRAC, hard=IMAGE_HARD, name=""):
    st = parity_report.image_stats(got, want)
    parity_report.record("image", f"{_where()} {name}".strip(), frac_allowed=frac, hard_allowed=hard, **st)
    assert np.isfinite(got).all()
    assert st["max_abs"] <= hard, f"max image diff {st['max_abs']}"
    assert st["frac_gt_1e4"] <= frac, f"{st['frac_gt_1e4']:.2e} of pixels differ by more than 1e-4"


def check_grad(got, want, name, rtol=1e-4, outlier_frac=GRAD_OUTLIERS):
    scale = max(1.0, float(np.abs(want).max())) if want.size else 1.0
    diff = np.abs(got - want)
    bad = diff > rtol * np.maximum(1.0, np.abs(want)) + 1e-5 * scale
    parity_report.record("grad", f"{_where()} {name}", frac_bad=float(bad.mean()) if bad.size else 0.0,
                         max_abs=float(diff.max()) if diff.size else 0.0, ref_max=scale, rtol=rtol,
                         frac_allowed=outlier_frac, n=int(diff.size),
                         max_rel=float((diff / np.maximum(1.0, np.abs(want))).max()) if diff.size else 0.0)
    assert np.isfinite(got).all(), name
    assert bad.mean() <= outlier_frac, f"{name}: {bad.mean():.2e} of entries off (max diff {diff.max():.3e})"
