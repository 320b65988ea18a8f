"""Tolerance checks of the GPU parity tests. Every call records what it measured (tests/parity_report.py).

Tolerances (fp32 path; BASELINE north_star: per-pixel match within 1e-4, gradients 1e-4 relative):
  * image: |diff| <= 1e-4 on all but `frac` (1e-6) of the pixels -- on the small frames of the suite (< 1 M values) that is NO value
    beyond 1e-4 -- and <= `hard` everywhere (one flipped (pixel, Gaussian) pair at the support cutoff: 0.99 e^-4.5 = 1.1e-2).
    WHOLE 1080p FRAMES (tests/test_full_size_parity.py, 6.2 M values, ~10^10 (pixel, Gaussian) pairs) get a COUNT budget per
    workload (FRAME_COUNT_BUDGETS there: S-6M 12 values, measured <= 9; S-6M-T 36, measured <= 28). The blend thresholds
    (alpha < 1/255, T < 1e-4, power < -4.5, power > 0) are discontinuous, and the HIP kernel uses the hardware exp2 and fused
    multiply-adds while the oracle follows the reference's literal fp32 expression, so a pair that sits within an ulp of a
    threshold may flip; each flip moves a pixel by at most ~alpha x colour x T: <= 1.1e-2 at the -4.5 cutoff, <= 3.9e-3 at
    alpha < 1/255. The measured count and maximum of every comparison are kept in tests/parity_report_gpu.json.
  * gradients: RELATIVE, row by row. A row is one Gaussian's slice of the tensor ([3] of dL_dmean3D, [16,3] of dL_dsh,
    a scalar of dL_dopacity ...): ||got_i - want_i||_2 <= rtol * ||want_i||_2 + floor_i on all but `outlier_frac` of the
    rows that carry a gradient. There is no floor tied to the tensor's LARGEST entry: floor_i is
        GRAD_FLOOR (1e-7) * the median row norm of the tensor -- fp32 rounding of a row far below the typical one --
      + `cancel` * scale_i when the caller passes per-row summand scales (`row_scale`): a sum of float atomics has no
        defined order in the reference either (the oracle sums in double), and a row whose terms cancel carries the
        rounding of its largest summand, not of its result.
    Budgets: small frames GRAD_OUTLIERS (3e-3) of the rows outside rtol and GRAD_GROSS (2e-4) outside 100 rtol (a few hundred
    rows, a flipped pair among them); WHOLE FRAMES one budget PER TENSOR from the measured worst x 1.25
    (FULL_FRAME_GRAD_BUDGETS: 3e-5 ... 2e-4, dL_dopacity 1.7e-3 / 2.9e-3) with the loss gradient zero on the handful of pixels
    whose FORWARD state differs between the two passes (a flipped pair changes its pixel's transmittance for every entry behind
    it: on S-6M-T's lists, thousands of entries deep, two dozen such pixels alone put 1.4e-3 of 1 M rows outside 1e-4 -- round 4's
    "1.7e-3 dL_dopacity outliers" were mostly that). What is left is arithmetic: the backward pass recovers T by dividing
    T_final by (1 - alpha) down the list (backward.cu:503-507) and forms dL/dalpha from differences (:516-521), so a row
    whose terms nearly cancel carries the rounding of its summands -- in the reference as here. check_against_noise() holds the
    HIP path to that yardstick: both the HIP gradients and the fp32 restatement of the reference are compared with the SAME
    arithmetic in double on the same forward state, and the HIP path may leave 1.5 x the reference's own share of rows outside
    1e-4 (measured 0.98-1.19 x). On top of the row test the tensor as a whole must agree: cosine similarity >= 1 - GRAD_COSINE and
    relative L2 error <= GRAD_REL_L2 (whole frames: 2e-5). A systematic error -- 1 % in the degree-3 SH gradients, say -- puts
    nearly every row outside rtol (tested).
"""
import os

import numpy as np

from tests import parity_report

# Bounds a few times above the worst case measured over the whole GPU suite (tests/parity_report_gpu.json):
IMAGE_FRAC = 1e-6   # share of pixels allowed above 1e-4
IMAGE_HARD = 1.11e-2 # ... and how far those may be off: one flipped (pixel, Gaussian) pair at the support cutoff (0.99 e^-4.5)
GRAD_OUTLIERS = 3e-3  # share of gradient rows allowed outside rtol (cancelling rows, threshold flips)
GRAD_GROSS = 2e-4     # ... and outside 100 x rtol
GRAD_FLOOR = 1e-7     # x median row norm
GRAD_COSINE = 1e-7    # 1 - cosine similarity of the whole tensor
GRAD_REL_L2 = 1e-3    # ||got - want|| / ||want|| of the whole tensor (dominated by the few flipped pairs)


def _where():
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::")[-1]


FLIP_BOUND = 0.99 * float(np.exp(-4.5)) * 1.001  # one (pixel, Gaussian) pair flipped at the support cutoff: alpha <= 0.99 e^-4.5, colour, T <= 1


def check_image(got, want, frac=IMAGE_FRAC, hard=IMAGE_HARD, name="", count=None):
    """count: a COUNT budget instead of the fraction (whole 1080p frames: 6.2 M values): at most `count` values beyond 1e-4, each
    no further off than one flipped (pixel, Gaussian) pair at the support cutoff can move a pixel (FLIP_BOUND = 0.99 e^-4.5 =
    1.1e-2; measured: <= 9 values = 3 pixels per frame, largest 4.9e-3)."""
    st = parity_report.image_stats(got, want)
    if count is not None:
        hard = FLIP_BOUND
        st["count_gt_1e4"] = int(round(st["frac_gt_1e4"] * st["n"]))
    parity_report.record("image", f"{_where()} {name}".strip(), frac_allowed=frac, hard_allowed=hard, count_allowed=count, **st)
    assert np.isfinite(got).all()
    assert st["max_abs"] <= hard, f"max image diff {st['max_abs']}"
    if count is not None:
        assert st["count_gt_1e4"] <= count, f"{st['count_gt_1e4']} values differ by more than 1e-4 (budget {count})"
    else:
        assert st["frac_gt_1e4"] <= frac, f"{st['frac_gt_1e4']:.2e} of pixels differ by more than 1e-4"


def grad_stats(got, want, rtol=1e-4, row_scale=None, cancel=4e-6):
    """Row-relative comparison of two gradient tensors (first axis = Gaussian). -> dict of what was measured."""
    g = np.asarray(got, np.float64).reshape(len(got), -1) if np.ndim(got) else np.asarray(got, np.float64).reshape(1, -1)
    w = np.asarray(want, np.float64).reshape(g.shape)
    d = g - w
    rn = np.sqrt((w * w).sum(axis=1))
    dn = np.sqrt((d * d).sum(axis=1))
    live = (rn > 0) | (dn > 0)
    n_live = int(live.sum())
    typ = float(np.median(rn[rn > 0])) if (rn > 0).any() else 0.0
    floor = GRAD_FLOOR * typ + (cancel * np.asarray(row_scale, np.float64).reshape(-1) if row_scale is not None else 0.0)
    bad = live & (dn > rtol * rn + floor)
    gross = live & (dn > 100.0 * rtol * rn + floor)
    rel = dn[live] / np.maximum(rn[live], 1e-300)
    wn, gn = float(np.sqrt((w * w).sum())), float(np.sqrt((g * g).sum()))
    cos = float((g * w).sum() / (wn * gn)) if wn > 0 and gn > 0 else 1.0
    return dict(rows=int(len(g)), rows_with_gradient=n_live, frac_bad=float(bad.sum() / max(n_live, 1)), frac_gross=float(gross.sum() / max(n_live, 1)),
                max_abs=float(np.abs(d).max()) if d.size else 0.0, ref_max=float(np.abs(w).max()) if w.size else 0.0,
                median_row_norm=typ, rel_l2=float(np.sqrt((d * d).sum()) / wn) if wn > 0 else 0.0, one_minus_cosine=1.0 - cos,
                row_rel_p50=float(np.quantile(rel, 0.5)) if n_live else 0.0, row_rel_p99=float(np.quantile(rel, 0.99)) if n_live else 0.0,
                row_rel_p9999=float(np.quantile(rel, 0.9999)) if n_live else 0.0, row_rel_max=float(rel.max()) if n_live else 0.0)


# Per-tensor outlier budgets of the WHOLE-FRAME comparisons (share of the rows with a gradient outside 1e-4 relative), from the measured
# worst x 1.25 (tests/parity_report_gpu.json; round 4 held every tensor to one 3e-3, 1.6 x above the worst of them and 7 x above the
# others): S-6M = 131 k rows, S-6M-T = 1.03 M rows walked thousands of entries deep. The budgets of S-6M-T are larger because the
# REFERENCE's arithmetic is noisier there: T is recovered by division down lists of thousands of faint entries and dL/dalpha is a
# difference of accumulated colours (R0 backward.cu:503-521); tests/test_full_size_parity.py measures the fp32 restatement of the
# reference against the same arithmetic in double on the same frame and holds the HIP path to that noise as well (check_against_noise).
FULL_FRAME_GRAD_BUDGETS = {
    # measured (round 5, pixels whose forward state differs left out of the loss, see _training_step):
    #   S-6M   131 k rows: opacity 1.34e-3, scale 1.5e-4, cov3D 1.5e-4, rot 8.4e-5, mean2D / sh 1.5e-5, mean3D 7.6e-6, colour 0, contributions 3.0e-5
    #   S-6M-T 1.03 M rows: opacity 2.34e-3, mean2D 4.9e-5, rot 4.6e-5, scale 3.3e-5, mean3D 3.0e-5, cov3D 2.0e-5, sh 9.8e-6, colour 0, contributions 8.6e-5
    # budget = measured x 1.25, and at least measured + 3e-5 (the float atomics' order moves a handful of rows from run to run)
    "S-6M": {"dL_dopacity": 1.7e-3, "dL_dmean2D": 5e-5, "dL_dcolor": 3e-5, "dL_dmean3D": 4e-5, "dL_dcov3D": 1.9e-4, "dL_dsh": 5e-5, "dL_dscale": 1.9e-4,
             "dL_drot": 1.2e-4, "contributions": 6e-5},
    # ring views 2 / 5 / 7 of S-6M (round 6; BASELINE config 5's per-rank frames): measured worst of the three views -- opacity 1.58e-3, cov3D 1.91e-4,
    # scale 1.74e-4, rot 1.07e-4, contributions 9.2e-5 (view 7, behind its 14 flipped image values), mean2D / mean3D 3.8e-5, sh / colour 7.6e-6 -- x 1.25,
    # and at least measured + 3e-5 (the float atomics' order moves a handful of rows from run to run)
    "S-6M ring": {"dL_dopacity": 2.0e-3, "dL_dmean2D": 7e-5, "dL_dcolor": 4e-5, "dL_dmean3D": 7e-5, "dL_dcov3D": 2.4e-4, "dL_dsh": 4e-5, "dL_dscale": 2.2e-4,
                  "dL_drot": 1.4e-4, "contributions": 1.2e-4},
    "S-6M-T": {"dL_dopacity": 2.9e-3, "dL_dmean2D": 8e-5, "dL_dcolor": 3e-5, "dL_dmean3D": 6e-5, "dL_dcov3D": 5e-5, "dL_dsh": 4e-5, "dL_dscale": 7e-5,
               "dL_drot": 8e-5, "contributions": 1.2e-4},
}
FULL_FRAME_GRAD_REL_L2 = 2e-5   # whole-tensor relative L2 error of a whole-frame gradient (measured <= 5.2e-6; the default 1e-3 is for frames with flips)
FULL_FRAME_GRAD_GROSS = 6e-5    # rows outside 100 x rtol (measured <= 3.3e-5, dL_dopacity)


def check_against_noise(got, f32, f64, name, rtol=1e-4, factor=1.5, floor=1e-4):
    """The HIP gradients `got` and the fp32 restatement of the reference `f32`, both against the same arithmetic in DOUBLE on the same
    forward state (`f64`): the HIP path may leave at most `factor` x the reference arithmetic's own share of rows outside rtol (+ `floor`:
    rows next to the few (pixel, Gaussian) pairs whose discrete decisions differ between the two forward passes)."""
    a, b = grad_stats(got, f64, rtol=rtol), grad_stats(f32, f64, rtol=rtol)
    parity_report.record("noise", f"{_where()} {name}", hip_vs_f64_frac_bad=a["frac_bad"], ref_f32_vs_f64_frac_bad=b["frac_bad"],
                         hip_vs_f64_p99=a["row_rel_p99"], ref_f32_vs_f64_p99=b["row_rel_p99"], hip_vs_f64_rel_l2=a["rel_l2"], ref_f32_vs_f64_rel_l2=b["rel_l2"],
                         hip_vs_f64_gross=a["frac_gross"], ref_f32_vs_f64_gross=b["frac_gross"], rows=a["rows_with_gradient"])
    if os.environ.get("FOVRASTER_MEASURE_BUDGETS") == "1":
        return  # calibration run: record, do not judge
    assert a["frac_bad"] <= factor * b["frac_bad"] + floor, \
        f"{name}: {a['frac_bad']:.2e} of the rows outside {rtol:g} of the double-precision gradients; the reference's own fp32 arithmetic: {b['frac_bad']:.2e}"
    assert a["row_rel_p99"] <= max(2.0 * b["row_rel_p99"], 5e-6), f"{name}: p99 {a['row_rel_p99']:.2e} vs the reference arithmetic's {b['row_rel_p99']:.2e}"


def check_grad(got, want, name, rtol=1e-4, outlier_frac=GRAD_OUTLIERS, row_scale=None, cosine=GRAD_COSINE, rel_l2=GRAD_REL_L2, gross_frac=GRAD_GROSS):
    st = grad_stats(got, want, rtol=rtol, row_scale=row_scale)
    parity_report.record("grad", f"{_where()} {name}", rtol=rtol, frac_allowed=outlier_frac, gross_allowed=gross_frac, **st)
    assert np.isfinite(got).all(), name
    # (a tensor with a few hundred rows may hold three such rows whatever the fraction says)
    assert st["frac_bad"] <= max(outlier_frac, 3.5 / max(st["rows_with_gradient"], 1)), \
        f"{name}: {st['frac_bad']:.2e} of the rows outside {rtol:g} relative (max |diff| {st['max_abs']:.3e})"
    assert st["frac_gross"] <= max(gross_frac, 1.5 / max(st["rows_with_gradient"], 1)), f"{name}: {st['frac_gross']:.2e} of the rows outside {100 * rtol:g} relative"
    assert st["one_minus_cosine"] <= cosine, f"{name}: 1 - cosine = {st['one_minus_cosine']:.3e}"
    assert st["rel_l2"] <= rel_l2, f"{name}: relative L2 error {st['rel_l2']:.3e}"
