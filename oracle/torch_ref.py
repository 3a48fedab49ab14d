"""Dense pure-torch restatement of SURVEY.md Appendix A.2-A.4  --  TEST INFRASTRUCTURE ONLY.

Independent check of the hand-written analytic backward in raster_oracle.c: the forward below
is written with ordinary differentiable torch ops (every cull / threshold / sort decision taken
under no_grad and then treated as a constant, straight-through min(0.99, .), clamped-Jacobian
convention A.5(ii)), so torch.autograd yields the gradients the operator must return.
O(Npix * P) memory: small cases only.  Works in float32 or float64.

Conventions cited in raster_oracle.c (scene/cameras.py:54-58, utils/general_utils.py:78-110,
utils/sh_utils.py:57-112).
"""
import torch

TILE = 16
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
      0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
      -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def _eval_sh(deg, sh, d):
    """sh [P,M,3], d [P,3] unit directions -> [P,3] (utils/sh_utils.py:57-112)."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
               + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def rasterize(H, W, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix, sh_degree, campos,
              means3D, opacities, scales=None, rotations=None, cov3D_precomp=None, shs=None,
              colors_precomp=None):
    """Returns (color[3,H,W], radii[P], means2D_ndc[P,2]).  means2D_ndc is p_proj.xy: d(loss)/d
    (means2D_ndc) is the operator's means2D gradient (A.5 vi)."""
    dt = means3D.dtype
    V = viewmatrix.to(dt)
    M = projmatrix.to(dt)
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    t = (ph @ V)[:, :3]
    hom = ph @ M
    p_w = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * p_w[:, None]
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    if cov3D_precomp is not None:
        c6 = cov3D_precomp
        S = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4],
                         c6[:, 5]], 1).reshape(P, 3, 3)
    else:
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        Rm = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                          2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                          2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
        L = Rm * (scale_modifier * scales)[:, None, :]
        S = L @ L.transpose(1, 2)
    tz = t[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    clx = (txtz < -limx) | (txtz > limx)
    cly = (tytz < -limy) | (tytz > limy)
    tx = torch.where(clx, (txtz.clamp(-limx, limx) * tz).detach(), t[:, 0])  # A.5 (ii)
    ty = torch.where(cly, (tytz.clamp(-limy, limy) * tz).detach(), t[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).reshape(P, 2, 3)
    Wm = V[:3, :3].T  # view = Wm @ world
    T = J @ Wm
    cov = T @ S @ T.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    with torch.no_grad():
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        rad = torch.ceil(3 * torch.sqrt(lam))
        mx_ = ((ndc[:, 0] + 1) * W - 1) * 0.5
        my_ = ((ndc[:, 1] + 1) * H - 1) * 0.5
        gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
        rminx = torch.trunc((mx_ - rad) / 16).clamp(0, gx)
        rminy = torch.trunc((my_ - rad) / 16).clamp(0, gy)
        rmaxx = torch.trunc((mx_ + rad + 15) / 16).clamp(0, gx)
        rmaxy = torch.trunc((my_ + rad + 15) / 16).clamp(0, gy)
        vis = (tz > 0.2) & (det != 0) & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
        radii = torch.where(vis, rad, torch.zeros_like(rad)).to(torch.int32)
        order = torch.argsort(tz.to(torch.float32), stable=True)  # depth, ties by index
    mx = ((ndc[:, 0] + 1) * W - 1) * 0.5
    my = ((ndc[:, 1] + 1) * H - 1) * 0.5
    Qxx, Qxy, Qyy = c / det, -b / det, a / det
    if colors_precomp is not None:
        col = colors_precomp
    else:
        d = means3D - campos.to(dt)[None]
        d = d / d.norm(dim=1, keepdim=True)
        col = torch.clamp(_eval_sh(sh_degree, shs, d) + 0.5, min=0.0)
    o = opacities.reshape(-1)

    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pxs, pys = xs.reshape(-1).to(dt), ys.reshape(-1).to(dt)
    tix, tiy = (xs.reshape(-1) // TILE).to(dt), (ys.reshape(-1) // TILE).to(dt)
    idx = order
    dx = mx[idx][None, :] - pxs[:, None]
    dy = my[idx][None, :] - pys[:, None]
    power = -0.5 * (Qxx[idx][None] * dx * dx + Qyy[idx][None] * dy * dy) - Qxy[idx][None] * dx * dy
    G = torch.exp(torch.clamp(power, max=0.0))
    al_raw = o[idx][None] * G
    alpha = al_raw + (torch.clamp(al_raw, max=0.99) - al_raw).detach()  # straight-through
    with torch.no_grad():
        in_rect = ((tix[:, None] >= rminx[idx][None]) & (tix[:, None] < rmaxx[idx][None]) &
                   (tiy[:, None] >= rminy[idx][None]) & (tiy[:, None] < rmaxy[idx][None]) & vis[idx][None])
        valid = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0)
        a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
        Tin = torch.cumprod(torch.cat([torch.ones(a_eff.shape[0], 1, dtype=dt), 1 - a_eff[:, :-1]], 1), 1)
        # the stop decision must see T as it would be *with* the stop, which equals Tin until
        # the first stop; splats after the first stop are excluded anyway
        stop = valid & (Tin * (1 - a_eff) < 1e-4)
        contrib = valid & (torch.cumsum(stop.to(torch.int32), 1) == 0)
    a_c = torch.where(contrib, alpha, torch.zeros_like(alpha))
    Tk = torch.cumprod(torch.cat([torch.ones(a_c.shape[0], 1, dtype=dt), 1 - a_c[:, :-1]], 1), 1)
    wgt = a_c * Tk
    Tfin = Tk[:, -1] * (1 - a_c[:, -1])
    img = wgt @ col[idx] + Tfin[:, None] * bg.to(dt)[None]
    return img.T.reshape(3, H, W), radii, ndc
