"""`lara_amd.loss.ms_ssim`: the reference's `pytorch_msssim.MS_SSIM(data_range=1.0, size_average=True, channel=3)`
(lightning/loss.py:15, :41-45).  The package is absent from this image and un-pinned in the reference (no requirements
file names it), so parity with the package itself is UNPINNED; what is checked: an independent float64 numpy restatement
of the published algorithm (explicit loops over the window taps, no shared code), and the properties the definition gives."""
import numpy as np
import pytest
import torch

from lara_amd.loss import MS_SSIM_WEIGHTS, ms_ssim, ms_ssim_terms


def _blur64(x, g):                       # x [H,W] float64, 'valid' separable filter, rows then columns
    k = len(g)
    a = sum(g[i] * x[i:x.shape[0] - k + 1 + i, :] for i in range(k))
    return sum(g[i] * a[:, i:a.shape[1] - k + 1 + i] for i in range(k))


def _ms_ssim64(X, Y):
    c = np.arange(11, dtype=np.float64) - 5
    g = np.exp(-c ** 2 / (2 * 1.5 ** 2)); g /= g.sum()
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    N, C = X.shape[:2]
    out = np.ones((N, C))
    for n in range(N):
        for ch in range(C):
            x, y = X[n, ch].astype(np.float64), Y[n, ch].astype(np.float64)
            for lvl, w in enumerate(MS_SSIM_WEIGHTS):
                mu1, mu2 = _blur64(x, g), _blur64(y, g)
                s1, s2, s12 = _blur64(x * x, g) - mu1 ** 2, _blur64(y * y, g) - mu2 ** 2, _blur64(x * y, g) - mu1 * mu2
                cs = (2 * s12 + C2) / (s1 + s2 + C2)
                ss = (2 * mu1 * mu2 + C1) / (mu1 ** 2 + mu2 ** 2 + C1) * cs
                v = ss.mean() if lvl == 4 else cs.mean()
                out[n, ch] *= max(v, 0.0) ** w
                if lvl < 4:
                    def pool(z):
                        H, W = z.shape
                        zp = np.zeros((H + H % 2 * 2, W + W % 2 * 2))       # avg_pool2d(padding = side % 2), zeros count
                        zp[H % 2:H % 2 + H, W % 2:W % 2 + W] = z
                        Ho, Wo = zp.shape[0] // 2, zp.shape[1] // 2
                        return zp[:2 * Ho, :2 * Wo].reshape(Ho, 2, Wo, 2).mean((1, 3))
                    x, y = pool(x), pool(y)
    return out.mean()


@pytest.mark.parametrize("H,W", [(176, 200), (191, 163)])
def test_ms_ssim_matches_an_independent_float64_restatement(H, W):
    g = torch.Generator().manual_seed(H)
    X = torch.rand(2, 3, H, W, generator=g)
    Y = (X + 0.1 * torch.randn(2, 3, H, W, generator=g)).clamp(0, 1)
    got = float(ms_ssim(X, Y))
    want = _ms_ssim64(X.numpy(), Y.numpy())
    assert 0.3 < want < 0.999
    assert got == pytest.approx(want, abs=2e-5)


def test_ms_ssim_properties_and_gradient():
    g = torch.Generator().manual_seed(0)
    X = torch.rand(1, 3, 192, 176, generator=g)
    Y = torch.rand(1, 3, 192, 176, generator=g)
    assert float(ms_ssim(X, X)) == pytest.approx(1.0, abs=1e-6)
    assert float(ms_ssim(X, Y)) == pytest.approx(float(ms_ssim(Y, X)), abs=1e-6)
    assert float(ms_ssim(X, Y)) < 0.2
    with pytest.raises(ValueError):
        ms_ssim(X[..., :160], Y[..., :160])
    Xd, Yd = X.double().requires_grad_(True), Y.double()     # autograd through the torch operators, against central differences
    ms_ssim(Xd, Yd).backward()
    i = (0, 1, 100, 90)
    e = 1e-5
    Xp, Xm = X.double(), X.double()
    Xp[i] += e; Xm[i] -= e
    fd = (float(ms_ssim(Xp, Yd)) - float(ms_ssim(Xm, Yd))) / (2 * e)
    assert float(Xd.grad[i]) == pytest.approx(fd, rel=1e-4)


def test_loss_with_the_ms_ssim_term_and_the_references_statistics():
    """`pipeline.lara_loss` (torch, CPU-able): loss = pixel terms + 0.5 (1 - MS_SSIM) per image (loss.py:45), statistics carry
    the reference's keys (loss.py:38-43, :49, :57)."""
    from lara_amd.pipeline import lara_loss
    g = torch.Generator().manual_seed(3)
    B, V, H, W = 1, 2, 176, 96
    batch = {"tar_rgb": torch.rand(B, V, H, W, 3, generator=g)}
    out = {k: torch.rand(B, H, V * W, c, generator=g).squeeze(-1) if c == 1 else torch.rand(B, H, V * W, c, generator=g)
           for k, c in (("image", 3), ("image_fine", 3), ("acc_map", 1), ("acc_map_fine", 1), ("rend_dist", 1),
                        ("rend_normal", 3), ("depth_normal", 3))}
    base, st0 = lara_loss(batch, out, 2000, ms_ssim=False)
    full, st1 = lara_loss(batch, out, 2000)
    terms, st = ms_ssim_terms(batch, out)
    assert float(full) == pytest.approx(float(base) + float(terms[""]) + float(terms["_fine"]), rel=1e-6)
    assert set(st1) == {"mse", "psnr", "ssim", "mse_fine", "psnr_fine", "ssim_fine", "distortion", "normal"}
    assert set(st0) == set(st1) - {"ssim", "ssim_fine"}
    assert float(st1["psnr"]) == pytest.approx(-10 * np.log10(float(st1["mse"])), rel=1e-6)
