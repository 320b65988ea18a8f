"""A naive PyTorch-CPU forward rasterizer of the reference's `diff-gaussian-rasterization` ("original", R0) and of its
`_pcheck_obb_sum` flavour's blend rule (RS: the extra power < -4.5 cutoff) -- BASELINE config 1 / SURVEY.md 7 step 1 as written.

TEST INFRASTRUCTURE ONLY (lives under tests/): an INDEPENDENT derivation of the gradients -- torch.autograd through a plain,
vectorised restatement of the forward pass -- beside the hand-written backward of the C oracle (oracle/fovraster_oracle.c,
which follows R0/cuda_rasterizer/backward.cu line by line) and its finite-difference probes. Never imported by the product
package, never on a GPU path. Written from the reference's forward files only:
  projection / covariance / conic / radius / rect     R0/cuda_rasterizer/forward.cu:74-262, auxiliary.h:41-56,139-164
  SH colour, clamped at zero                           forward.cu:20-71
  keys / order: (tile, depth), stable in index         rasterizer_impl.cu:70-111, 300-308
  blend                                                forward.cu:331-382 (RS: :376-380 power < -4.5 also skipped)
All discrete decisions (cull, tile rectangles, order, skip tests, the T < 1e-4 stop) are evaluated without gradient, exactly as
in the reference, where they are branches; the differentiable path is what the reference's backward differentiates.
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)


def sh_colour(deg, sh, dirs):
    """forward.cu:20-71: sh [P,16,3], dirs [P,3] (unnormalised) -> rgb [P,3] clamped at 0."""
    d = dirs / dirs.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def rasterize(means3D, scales, rotations, opacities, shs, cam, cutoff=False):
    """-> image [3,H,W] (float64 in, float64 out). cam: the oracle's camera dict (tests/helpers.cam_dict).
    cutoff: RS / RP / RF's extra skip of power < -4.5."""
    f64 = torch.float64
    W, H = int(cam["image_width"]), int(cam["image_height"])
    vm = torch.as_tensor(cam["viewmatrix"], dtype=f64).reshape(4, 4)   # row-major memory of the TRANSPOSED matrix: p_view = p @ vm
    pm = torch.as_tensor(cam["projmatrix"], dtype=f64).reshape(4, 4)
    campos = torch.as_tensor(cam["campos"], dtype=f64)
    bg = torch.as_tensor(cam["bg"], dtype=f64)
    tanx, tany = float(cam["tanfovx"]), float(cam["tanfovy"])
    fx, fy = W / (2.0 * tanx), H / (2.0 * tany)
    mod = float(cam.get("scale_modifier", 1.0))
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=f64)
    ph = torch.cat([means3D, ones], 1) @ pm                          # auxiliary.h:58-77 (matrix[0],[4],[8],[12] = first component)
    pw = 1.0 / (ph[:, 3] + 1e-7)
    proj = ph[:, :3] * pw[:, None]
    t = torch.cat([means3D, ones], 1) @ vm
    tz = t[:, 2]
    in_front = tz > 0.2                                               # auxiliary.h:154
    # 3D covariance: Sigma = R S^2 R^T with the quaternion AS GIVEN (forward.cu:118-152)
    r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
    S = torch.diag_embed(mod * scales)
    Mm = R @ S
    Sigma = Mm @ Mm.transpose(1, 2)
    # EWA (forward.cu:74-113): clamped tangents, J, W = rotation part of the view matrix
    limx, limy = 1.3 * tanx, 1.3 * tany
    txc = torch.minimum(torch.full_like(tz, limx), torch.maximum(torch.full_like(tz, -limx), t[:, 0] / tz)) * tz
    tyc = torch.minimum(torch.full_like(tz, limy), torch.maximum(torch.full_like(tz, -limy), t[:, 1] / tz)) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * txc) / (tz * tz), zero, fy / tz, -(fy * tyc) / (tz * tz)], 1).reshape(P, 2, 3)
    Wv = vm[:3, :3].T                                                 # world -> camera rotation
    Tm = J @ Wv
    cov = Tm @ Sigma @ Tm.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = in_front & (det != 0)
    det_s = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([c / det_s, -b / det_s, a / det_s], 1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    pix = torch.stack([((proj[:, 0] + 1.0) * W - 1.0) * 0.5, ((proj[:, 1] + 1.0) * H - 1.0) * 0.5], 1)   # auxiliary.h:41-44
    gx, gy = (W + 15) // 16, (H + 15) // 16
    with torch.no_grad():                                             # getRect, auxiliary.h:46-56 (C truncation towards zero)
        x0 = torch.clamp(torch.trunc((pix[:, 0] - radius) / 16), 0, gx)
        x1 = torch.clamp(torch.trunc((pix[:, 0] + radius + 15) / 16), 0, gx)
        y0 = torch.clamp(torch.trunc((pix[:, 1] - radius) / 16), 0, gy)
        y1 = torch.clamp(torch.trunc((pix[:, 1] + radius + 15) / 16), 0, gy)
        ok = ok & ((x1 - x0) * (y1 - y0) > 0)
        order = torch.argsort(tz.masked_fill(~ok, float("inf")), stable=True)  # depth order, ties in index order
        order = order[: int(ok.sum())]
    rgb = sh_colour(int(cam["sh_degree"]), shs, means3D - campos)
    o = opacities.reshape(-1)
    # every pixel against every visible Gaussian in depth order (a P x pixels table: fine at 1k x 64k)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=f64), torch.arange(W, dtype=f64), indexing="ij")
    px, py = xs.reshape(-1), ys.reshape(-1)
    tx_pix, ty_pix = torch.div(px, 16, rounding_mode="floor"), torch.div(py, 16, rounding_mode="floor")
    g = order
    dx = pix[g, 0][:, None] - px[None, :]
    dy = pix[g, 1][:, None] - py[None, :]
    power = -0.5 * (conic[g, 0][:, None] * dx * dx + conic[g, 2][:, None] * dy * dy) - conic[g, 1][:, None] * dx * dy
    alpha = torch.clamp_max(o[g][:, None] * torch.exp(power), 0.99)
    with torch.no_grad():
        in_tile = (tx_pix[None, :] >= x0[g][:, None]) & (tx_pix[None, :] < x1[g][:, None]) & \
                  (ty_pix[None, :] >= y0[g][:, None]) & (ty_pix[None, :] < y1[g][:, None])
        keep = in_tile & ~(power > 0) & ~(alpha < 1.0 / 255.0)
        if cutoff:
            keep = keep & ~(power < -4.5)
    a_eff = torch.where(keep, alpha, torch.zeros_like(alpha))
    T_after = torch.cumprod(1.0 - a_eff, 0)
    T_before = torch.cat([torch.ones(1, T_after.shape[1], dtype=f64), T_after[:-1]], 0)
    with torch.no_grad():                                             # forward.cu:366-371: T' < 1e-4 ends the pixel, that Gaussian excluded
        dead = torch.cumsum((keep & (T_after < 1e-4)).to(torch.int64), 0) > 0
        contrib = keep & ~dead
        n_alive = (~dead).sum(0)                                      # entries processed before the stop
    w = torch.where(contrib, a_eff * T_before, torch.zeros_like(a_eff))
    colour = torch.einsum("gp,gc->cp", w, rgb[g])
    idx_last = torch.clamp(n_alive - 1, min=0)
    T_final = torch.where(n_alive > 0, T_after.gather(0, idx_last[None, :])[0], torch.ones_like(px))
    img = colour + T_final[None, :] * bg[:, None]
    return img.reshape(3, H, W)
