"""The MS-SSIM term on the GPU (`lara_amd.loss.ms_ssim_fused`, csrc/msssim.hip, include/lara_loss.h) against the torch formulation
`lara_amd.loss.ms_ssim` -- the restatement of `pytorch_msssim.MS_SSIM(data_range=1.0, size_average=True, channel=3)` (lightning/
loss.py:15, :42) that tests/test_loss_cpu.py holds to an independent float64 restatement.  The package itself is absent from this
image and un-pinned in the reference: PARITY WITH THE PACKAGE IS UNPINNED; these tests pin the kernels to the restatement."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(B, V, H, W, seed, noise=0.1):
    g = torch.Generator().manual_seed(seed)
    tar = torch.rand(B, V, H, W, 3, generator=g)
    side = tar.permute(0, 2, 1, 3, 4).reshape(B, H, V * W, 3)
    img = (side + noise * torch.randn(side.shape, generator=g)).clamp(0, 1)
    return img.to(DEV), tar.to(DEV)


def _torch_value_and_grad(img, tar, on_cpu_in_float64=False):
    """The torch formulation on the same tensors: on the GPU in fp32, or on the CPU in float64 (small cases)."""
    from lara_amd.loss import ms_ssim
    B, V, H, W = tar.shape[:4]
    if on_cpu_in_float64:
        img, tar = img.cpu().double(), tar.cpu().double()
    x = img.clone().requires_grad_(True)
    t = tar.permute(0, 2, 1, 3, 4).reshape(B, H, V * W, 3).permute(0, 3, 1, 2)
    val = ms_ssim(x.permute(0, 3, 1, 2), t)
    val.backward()
    return val.detach().float().to(DEV), x.grad.float().to(DEV)


@pytest.mark.parametrize("B,V,H,W", [(1, 2, 176, 96), (2, 1, 191, 163), (1, 3, 200, 67), (1, 1, 333, 170)])
def test_fused_ms_ssim_matches_the_torch_formulation(hip_lib, B, V, H, W):
    """Even and odd sides on every scale (the 2 x 2 pooling pads odd sides), several views side by side (the filter runs across
    the view boundaries of the stacked image, as in the reference), ragged tiles: value to 2e-6, gradient to 2e-4 of its maximum."""
    from lara_amd.loss import ms_ssim_fused
    img, tar = _pair(B, V, H, W, seed=H + W)
    want, gwant = _torch_value_and_grad(img, tar, on_cpu_in_float64=True)
    w32, g32 = _torch_value_and_grad(img, tar)
    print("torch fp32 on the GPU vs torch float64 on the CPU: value", float(w32 - want), "gradient (of max)",
          float((g32 - gwant).abs().max() / gwant.abs().max()))
    x = img.clone().requires_grad_(True)
    got = ms_ssim_fused(x, tar)
    got.backward()
    assert 0.2 < float(want) < 0.999
    assert float(got) == pytest.approx(float(want), abs=2e-6)
    scale = float(gwant.abs().max())
    assert float((x.grad - gwant).abs().max()) <= 2e-4 * scale, (float((x.grad - gwant).abs().max()), scale)
    # identical images: 1, and a random pair: small
    assert float(ms_ssim_fused(img, tar)) == pytest.approx(float(got), abs=0)
    same = tar.permute(0, 2, 1, 3, 4).reshape(B, H, V * W, 3).contiguous()
    assert float(ms_ssim_fused(same, tar)) == pytest.approx(1.0, abs=1e-6)


def test_fused_ms_ssim_at_training_size_and_inside_the_loss(hip_lib):
    """4 scenes x 8 views @512^2 (the step's images): value and gradient against the torch formulation, bit-repeatable; and
    `lara_loss(..., ms_ssim=True)` = pixel terms + 0.5 (1 - MS_SSIM) per image (loss.py:45) with the reference's statistics keys."""
    from lara_amd.loss import lara_loss, ms_ssim_fused
    img, tar = _pair(4, 8, 512, 512, seed=1, noise=0.2)
    want, gwant = _torch_value_and_grad(img, tar)
    runs = []
    for _ in range(2):
        x = img.clone().requires_grad_(True)
        v = ms_ssim_fused(x, tar)
        v.backward()
        runs.append((v.detach().clone(), x.grad.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert float(runs[0][0]) == pytest.approx(float(want), abs=2e-6)
    assert float((runs[0][1] - gwant).abs().max()) <= 2e-4 * float(gwant.abs().max())
    g = torch.Generator().manual_seed(3)
    out = {"image": img, "image_fine": (img + 0.05 * torch.randn(img.shape, generator=g).to(DEV)).clamp(0, 1),
           "acc_map_fine": torch.ones(img.shape[:3], device=DEV)}
    base, st0 = lara_loss({"tar_rgb": tar}, out, 500, ms_ssim=False)
    full, st1 = lara_loss({"tar_rgb": tar}, out, 500, ms_ssim=True)
    extra = 0.5 * (1 - ms_ssim_fused(out["image"], tar)) + 0.5 * (1 - ms_ssim_fused(out["image_fine"], tar))
    assert float(full) == pytest.approx(float(base) + float(extra), rel=1e-6)
    assert set(st1) - set(st0) == {"ssim", "ssim_fine"}


def test_fused_ms_ssim_rejects_small_images_and_cpu_tensors(hip_lib):
    from lara_amd.loss import ms_ssim_fused
    img, tar = _pair(1, 1, 160, 200, seed=0)
    with pytest.raises(ValueError):
        ms_ssim_fused(img, tar)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ms_ssim_fused(torch.rand(1, 176, 176, 3), torch.rand(1, 1, 176, 176, 3))
