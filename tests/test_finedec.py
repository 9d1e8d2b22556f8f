"""`Decoder.forward_fine` (lightning/network.py:280-284): the fused HIP path against the reference's own output and
autograd gradients (tests/golden/finedec_ref.npz, generated from the imported reference class by
tests/golden/make_finedec_fixture.py).  Tolerances: fp32 everywhere; the folded matrices re-associate the four
projections, so outputs agree to ~1e-6 relative -- bar: 2e-5 * max|ref| for the output and the input gradients,
1e-4 * max|ref| for the parameter gradients (sums over all points)."""
import os

import numpy as np
import pytest
import torch

from oracle.finedec_ref import FineDecoderRef, forward_fine_folded, hidden_preactivations

FX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "finedec_ref.npz")
PARAMS = ["norm.weight", "norm.bias", "cross_att.q_proj_weight", "cross_att.k_proj_weight", "cross_att.v_proj_weight",
          "cross_att.out_proj.weight", "mlp_fine.0.weight", "mlp_fine.0.bias", "mlp_fine.2.weight", "mlp_fine.2.bias"]


def _close(got, want, rel, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want).max()
    assert err <= rel * max(np.abs(want).max(), 1e-12), f"{what}: max err {err:.3e} vs max |ref| {np.abs(want).max():.3e}"


def test_oracle_modules_reproduce_the_reference_fixture():
    """The stand-in modules ARE the reference's (same torch classes, same call): bit-for-bit on the same CPU build,
    to rounding elsewhere."""
    fx = np.load(FX)
    dec = FineDecoderRef.from_fixture(fx)
    vol = torch.from_numpy(fx["vol"]).requires_grad_(True)
    pfp = torch.from_numpy(fx["pf_planar"]).requires_grad_(True)
    sh = dec.forward_fine(vol, torch.einsum('lcb->blc', pfp))
    (sh * torch.from_numpy(fx["gout"])).sum().backward()
    _close(sh.detach().numpy(), fx["sh"], 1e-6, "sh")
    _close(vol.grad.numpy(), fx["d_vol"], 1e-5, "d_vol")
    _close(pfp.grad.numpy(), fx["d_pf_planar"], 1e-5, "d_pf")


def test_folded_algebra_matches_the_reference_fixture():
    """The folding of include/lara_finedec.h (Wqk, W1ov) is exact algebra: output and every gradient -- including the
    reference's own unfolded parameters, reached through `_fold_fine_weights` -- match the fixture."""
    from lara_amd.fine import _fold_fine_weights
    fx = np.load(FX)
    dec = FineDecoderRef.from_fixture(fx)
    vol = torch.from_numpy(fx["vol"]).requires_grad_(True)
    pfp = torch.from_numpy(fx["pf_planar"]).requires_grad_(True)
    xn = torch.nn.functional.layer_norm(vol, (80,), dec.norm.weight, dec.norm.bias, dec.norm.eps)
    sh = forward_fine_folded(xn, pfp, *_fold_fine_weights(dec)).unsqueeze(1)
    (sh * torch.from_numpy(fx["gout"])).sum().backward()
    _close(sh.detach().numpy(), fx["sh"], 2e-5, "sh")
    _close(vol.grad.numpy(), fx["d_vol"], 2e-5, "d_vol")
    _close(pfp.grad.numpy(), fx["d_pf_planar"], 2e-5, "d_pf")
    params = dict(dec.named_parameters())
    for k in PARAMS:
        _close(params[k].grad.numpy(), fx["g." + k], 1e-4, "grad " + k)


def test_fold_refuses_other_sizes():
    from lara_amd.fine import _fold_fine_weights
    dec = FineDecoderRef()
    dec.cross_att = torch.nn.MultiheadAttention(80, 4, kdim=8, vdim=8, bias=False, batch_first=True)
    with pytest.raises(RuntimeError, match="LaRa's sizes"):
        _fold_fine_weights(dec)
    # sh_degree != 1 (3 or 27 outputs): the kernel stores 12 floats per point, so the wrapper must refuse
    with pytest.raises(RuntimeError, match="LaRa's sizes"):
        _fold_fine_weights(FineDecoderRef(sh_dim=3))


@pytest.mark.gpu
def test_hip_forward_fine_matches_the_reference_fixture():
    from lara_amd.fine import forward_fine
    fx = np.load(FX)
    dec = FineDecoderRef.from_fixture(fx).cuda()
    vol = torch.from_numpy(fx["vol"]).cuda().requires_grad_(True)
    pfp = torch.from_numpy(fx["pf_planar"]).cuda().requires_grad_(True)
    sh = forward_fine(dec, vol, torch.einsum('lcb->blc', pfp))
    assert sh.shape == (vol.shape[0], 1, 12) and sh.dtype == torch.float32
    (sh * torch.from_numpy(fx["gout"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    _close(sh.detach().cpu().numpy(), fx["sh"], 2e-5, "sh")
    _close(vol.grad.cpu().numpy(), fx["d_vol"], 2e-5, "d_vol")
    _close(pfp.grad.cpu().numpy(), fx["d_pf_planar"], 2e-5, "d_pf")
    params = dict(dec.named_parameters())
    for k in PARAMS:
        _close(params[k].grad.cpu().numpy(), fx["g." + k], 1e-4, "grad " + k)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 255, 256, 257, 100_003])
def test_hip_forward_fine_vs_oracle_ragged_sizes(n):
    """Sizes around the 256-point tile and one that takes several trips of the persistent workgroups."""
    from lara_amd.fine import _FineDecoder, _fold_fine_weights
    torch.manual_seed(n)
    dec = FineDecoderRef()
    with torch.no_grad():
        for p in dec.parameters():
            p.add_(torch.randn_like(p) * 0.1)
    W = [w.detach() for w in _fold_fine_weights(dec)]
    xn = torch.randn(n, 80)
    pf = torch.randn(4, 8, n)
    gout = torch.randn(n, 12)
    # A point with a hidden unit within 1e-5 of ReLU's kink may take either branch in fp32 (the two sides sum their 64 / 80
    # products in different orders), which changes its gradients -- and the weight gradients -- by O(1).  64 units per
    # point with O(1) spread put ~2e-3 of random points there: those are nudged off the kink before either side runs.
    for _ in range(20):
        on_kink = hidden_preactivations(xn, pf, W[0], W[1], W[2]).abs().min(dim=1).values < 1e-5
        if not on_kink.any():
            break
        xn[on_kink] = xn[on_kink] * 1.003 + 0.001
    assert not on_kink.any()
    a = [t.clone().requires_grad_(True) for t in (xn, pf, *W)]
    ref = forward_fine_folded(*a)
    (ref * gout).sum().backward()
    b = [t.clone().cuda().requires_grad_(True) for t in (xn, pf, *W)]
    got = _FineDecoder.apply(*b)
    (got * gout.cuda()).sum().backward()
    torch.cuda.synchronize()
    _close(got.detach().cpu().numpy(), ref.detach().numpy(), 2e-5, "sh")
    for x, y, name in zip(b, a, ("xn", "pf", "Wqk", "W1ov", "b1", "W2", "b2")):
        _close(x.grad.cpu().numpy(), y.grad.numpy(), 2e-4 if name[0] in "Wb" else 2e-5, "grad " + name)


@pytest.mark.gpu
def test_hip_forward_fine_empty_and_cpu_inputs():
    from lara_amd.fine import _FineDecoder, _fold_fine_weights, forward_fine
    dec = FineDecoderRef().cuda()
    sh = forward_fine(dec, torch.zeros(0, 80, device="cuda"), torch.zeros(0, 4, 8, device="cuda"))
    assert sh.shape == (0, 1, 12)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _FineDecoder.apply(torch.zeros(3, 80), torch.zeros(4, 8, 3), *[w.detach().cpu() for w in _fold_fine_weights(dec)])


@pytest.mark.gpu
def test_take_rows_matches_advanced_indexing():
    from lara_amd.fine import take_rows
    torch.manual_seed(0)
    x = torch.randn(1000, 4, 3, device="cuda", requires_grad=True)
    y = x.detach().clone().requires_grad_(True)
    idx = (torch.rand(1000, device="cuda") > 0.4).nonzero().squeeze(-1)
    g = torch.randn(idx.numel(), 4, 3, device="cuda")
    a, b = take_rows(x, idx), y[idx]
    assert torch.equal(a, b)
    a.backward(g)
    b.backward(g)
    assert torch.equal(x.grad, y.grad)


@pytest.mark.gpu
def test_take_rows_multi_matches_advanced_indexing():
    """The five `x[mask]` of network.py:514-524 as one launch per direction (`lara_take_rows`): values and gradients bit for bit
    those of advanced indexing; an output nobody uses gets no gradient; an empty subset works."""
    from lara_amd.fine import take_rows_multi
    torch.manual_seed(1)
    shapes = [(3,), (4, 3), (1,), (2,), (4,)]
    xs = [torch.randn(3001, *s, device="cuda", requires_grad=True) for s in shapes]
    ys = [x.detach().clone().requires_grad_(True) for x in xs]
    idx = (torch.rand(3001, device="cuda") > 0.47).nonzero().squeeze(-1)
    outs = take_rows_multi(xs, idx)
    refs = [y[idx] for y in ys]
    for a, b in zip(outs, refs):
        assert a.shape == b.shape and torch.equal(a, b)
    gs = [torch.randn_like(b) for b in refs]
    use = [0, 1, 3, 4]                       # the opacity output's gradient stays undefined
    torch.autograd.backward([outs[k] for k in use], [gs[k] for k in use])
    torch.autograd.backward([refs[k] for k in use], [gs[k] for k in use])
    for k, (x, y) in enumerate(zip(xs, ys)):
        if k in use:
            assert torch.equal(x.grad, y.grad), k
        else:
            assert x.grad is None or not bool(x.grad.any()), k
    empty = take_rows_multi([x.detach() for x in xs], idx[:0])
    assert [tuple(e.shape) for e in empty] == [(0,) + s for s in shapes]
