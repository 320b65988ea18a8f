"""How far can nvcc's default multiply-add contraction move a whole frame?

Runs the CPU oracle's two float flavours -- f32 (-ffp-contract=off, the checker of every parity test) and f32_fma
(-ffp-contract=fast -mfma: every product that feeds an add / subtract fused, as -fmad=true allows; oracle/oracle.py FLAVOURS) -- over
whole 1080p frames of the S-6M and S-6M-T clouds (the bench's camera, ring view 0) and counts what differs (tests/envelope.py):
Gaussians whose radius changes, (tile, Gaussian) instances in one frame only, positions whose order changes, image values further
apart than 1e-4. TEST INFRASTRUCTURE: imports the oracle, never the product library; needs no GPU.

    python tools/fma_envelope.py [--out tests/fma_envelope_full.json] [--threads N] [--clouds S-6M,S-6M-T]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import cam_dict, scene_dict, syn  # noqa: E402
from tests.envelope import differences  # noqa: E402
from oracle import oracle as orc  # noqa: E402

W, H = 1920, 1080


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "fma_envelope_full.json"))
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--clouds", default="S-6M,S-6M-T")
    ap.add_argument("--P", type=int, default=6_000_000)
    a = ap.parse_args()
    orc.set_threads(a.threads)
    cam = syn.camera_ring(0, 8, W, H)
    frames = [("pcheck_obb_sum", (0.5, 0.5), "training frame"), ("fov_pcheck_obb", (0.5, 0.5), "foveated, centre gaze"),
              ("fov_pcheck_obb", (0.25, 0.75), "foveated, bench gaze 2")]
    doc = dict(what="oracle f32 (-ffp-contract=off) vs f32_fma (-ffp-contract=fast -mfma), whole 1080p frames", threads=a.threads, entries=[])
    for name in a.clouds.split(","):
        logit = syn.OPACITY_LOGIT_S6M if name == "S-6M" else syn.OPACITY_LOGIT_S6MT
        cloud = syn.scene_bicycle_scale(P=a.P, seed=1, opacity_logit=logit)
        fov = syn.foveation_layers(cloud, seed=2)
        scenes = {"pcheck_obb_sum": scene_dict(cloud, "pcheck_obb_sum"), "fov_pcheck_obb": scene_dict(cloud, "fov_pcheck_obb", fov)}
        del cloud
        for variant, gaze, label in frames:
            cd = cam_dict(cam, bg=(0.05, 0.1, 0.15), gaze=gaze, alpha=0.05)
            cd["capacity_hint"] = 20_000_000
            t0 = time.time()
            x = orc.forward(variant, scenes[variant], cd)
            y = orc.forward(variant, scenes[variant], cd, fma=True)
            d = differences(x, y)
            e = dict(cloud=name, variant=variant, frame=label, gaze=list(gaze), **d, seconds=round(time.time() - t0, 1))
            if variant == "pcheck_obb_sum":
                e["n_contrib_differ"] = int((x["n_contrib"] != y["n_contrib"]).sum())
                e["gaussians_count_differ"] = int((x["gaussians_count"] != y["gaussians_count"]).sum())
            print(json.dumps(e), flush=True)
            doc["entries"].append(e)
            del x, y
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
