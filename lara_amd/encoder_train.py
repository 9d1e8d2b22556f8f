"""Training path of LaRa's volume transformer on MI355X: forward AND backward on the HIP kernels of
``liblara2dgs.so`` (csrc/encoder.hip, csrc/encoder_bwd.hip).

``VolTransformer`` here is a drop-in for the reference module of the same name
(lightning/network.py:105-164): the same constructor arguments, the same ``forward(image_feats)``,
and -- because it is built from the same torch sub-modules, used purely as parameter containers -- the
same ``state_dict`` keys and fp32 master parameters, so the reference's optimiser, DDP wrapper and
checkpoints work on it unchanged.  What changes is what runs: ``forward`` casts the weights to bf16
(the precision the reference's matmuls run in under ``precision="bf16-mixed"``,
train_lightning.py:74) and launches the HIP forward, saving only each layer's input rows; ``backward``
re-runs each layer's forward inside ``lara_groupblock_backward`` and returns fp32 gradients for
every parameter and for ``image_feats``.

There is no CPU path and no torch fallback: tensors must live on the GPU and the library must load.
"""
from __future__ import annotations

import ctypes
import math

import torch
from torch import nn

from .encoder import _BlockWeights, _lib as _fwd_lib, tokens_to_volume, volume_to_tokens
from .rasterizer import _check

_configured = False


class _BlockWeightsT(ctypes.Structure):  # struct lara_groupblock_weights_t
    _fields_ = [(n, ctypes.c_void_p) for n in ("wq_t", "wkv_t", "wo_t", "w1_t", "w2_t", "wconv_t")]


_GRAD_FIELDS = ("ln1_w", "ln1_b", "wq", "wkv", "wo", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2", "ln3_w", "ln3_b", "wconv")


class _BlockGrads(ctypes.Structure):  # struct lara_groupblock_grads
    _fields_ = [(n, ctypes.c_void_p) for n in _GRAD_FIELDS]


def _lib():
    global _configured
    lib = _fwd_lib()
    if not _configured:
        vp, i32, f32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_int64
        lib.lara_groupblock_backward_workspace_bytes.restype = i64
        lib.lara_groupblock_backward_workspace_bytes.argtypes = [i32, i32]
        lib.lara_groupblock_backward.restype = ctypes.c_int
        lib.lara_groupblock_backward.argtypes = [i32, i32, i32, vp, vp, ctypes.POINTER(_BlockWeights),
                                                 ctypes.POINTER(_BlockWeightsT), vp, vp, vp, ctypes.POINTER(_BlockGrads), i32,
                                                 vp, i32, vp, vp]
        lib.lara_gemm_nt_bf16.restype = ctypes.c_int
        lib.lara_gemm_nt_bf16.argtypes = [i32, i32, i32, vp, vp, vp, i32, vp]
        lib.lara_batched_transpose.restype = ctypes.c_int
        lib.lara_batched_transpose.argtypes = [i32, i32, i32, vp, vp, i32, vp]
        lib.lara_groupblock_save_bytes.restype = i64
        lib.lara_groupblock_save_bytes.argtypes = [i32, i32]
        lib.lara_groupblock_forward_train.restype = ctypes.c_int
        lib.lara_groupblock_forward_train.argtypes = [i32, i32, i32, vp, vp, vp, ctypes.POINTER(_BlockWeights), vp, vp]
        lib.lara_voltrans_head_backward_workspace_bytes.restype = i64
        lib.lara_voltrans_head_backward_workspace_bytes.argtypes = [i32, i32, i32]
        lib.lara_voltrans_head_backward.restype = ctypes.c_int
        lib.lara_voltrans_head_backward.argtypes = [i32, i32, vp, vp, vp, f32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.lara_gemm_tn_workspace_bytes.restype = i64
        lib.lara_gemm_tn_workspace_bytes.argtypes = []
        lib.lara_gemm_tn_bf16.restype = ctypes.c_int
        lib.lara_gemm_tn_bf16.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp]
        lib.lara_layernorm256_backward.restype = ctypes.c_int
        lib.lara_layernorm256_backward.argtypes = [i32, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp]
        lib.lara_groupattn_core_backward.restype = ctypes.c_int
        lib.lara_groupattn_core_backward.argtypes = [i32, vp, vp, vp, vp, vp, vp]
        _configured = True
    return lib


def _keep_activations() -> bool:
    """Default: keep every block's intermediates in HBM between forward and backward (0.94 GB per layer at
    4 scenes x 32^3).  LARA_ENCODER_RECOMPUTE=1 saves only each block's input rows and re-runs its forward
    inside the backward instead (slower by the cost of one forward)."""
    import os
    return os.environ.get("LARA_ENCODER_RECOMPUTE", "0") != "1"


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


_scratch = {}   # (device index, tag) -> uint8 tensor, grown on demand


def _workspace(dev, tag: str, nbytes: int) -> torch.Tensor:
    if nbytes < 0:
        _check(int(nbytes), f"workspace size query ({tag})")
    key = (dev.index, tag)
    t = _scratch.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _scratch[key] = t
    return t


# per-layer parameters handed to the autograd function, in this order (reference layouts, fp32)
_LAYER_PARAMS = ("norm1.weight", "norm1.bias", "cross_attn.q_proj_weight", "cross_attn.k_proj_weight",
                 "cross_attn.v_proj_weight", "cross_attn.out_proj.weight", "norm2.weight", "norm2.bias",
                 "mlp.0.weight", "mlp.0.bias", "mlp.3.weight", "mlp.3.bias", "norm3.weight", "norm3.bias", "cnn.weight")
_NLP = len(_LAYER_PARAMS)


_wcache = {}   # (address, version) of a block's 15 parameters -> (the parameters, their bf16 / fp32 kernel operands)


def _layer_bf16(p):
    """fp32 reference-layout parameters of one block -> the kernels' operands (bf16 matrices in the layouts of
    lara_groupblock_weights, fp32 vectors).  Cached while the parameters are unchanged (same tensor objects, same
    versions): micro-batches of a gradient-accumulation step and evaluation loops reuse the casts."""
    key = tuple((t.data_ptr(), t._version) for t in p)   # (the cache keeps the tensors, hence their storage, alive)
    hit = _wcache.get(key)
    if hit is not None:
        return hit[1]
    (ln1w, ln1b, wq, wk, wv, wo, ln2w, ln2b, w1, b1, w2, b2, ln3w, ln3b, wc) = p
    bf = torch.bfloat16
    f = {"ln1_w": ln1w.float().contiguous(), "ln1_b": ln1b.float().contiguous(),
         "ln2_w": ln2w.float().contiguous(), "ln2_b": ln2b.float().contiguous(),
         "ln3_w": ln3w.float().contiguous(), "ln3_b": ln3b.float().contiguous(),
         "b1": b1.float().contiguous(), "b2": b2.float().contiguous(),
         "wq": wq.to(bf).contiguous(), "wkv": torch.cat([wk, wv], 0).to(bf).contiguous(), "wo": wo.to(bf).contiguous(),
         "w1": w1.to(bf).contiguous(), "w2": w2.to(bf).contiguous(),
         # cnn.weight [out, in, kd, kh, kw] -> [out][tap][in] (cast first: the re-layout then moves half the bytes)
         "wconv": wc.to(bf).permute(0, 2, 3, 4, 1).reshape(256, 27 * 256).contiguous()}
    if len(_wcache) >= 64:          # a few models' worth of blocks; stale entries (old versions) are dropped wholesale
        _wcache.clear()
    _wcache[key] = (list(p), f)
    return f


def invalidate_weight_cache():
    """Drop the cached bf16 operands.  Needed only after parameters were written through `.data` (e.g. an EMA
    swap-in `p.data.copy_(ema)`), which does not bump the version counter the cache is keyed on; optimiser steps,
    `copy_` under no_grad and `load_state_dict` do bump it."""
    _wcache.clear()


def _layer_bf16_t(f):
    """The transposed operands of the backward's dX GEMMs; kept with the entry of `_layer_bf16` they derive from."""
    t = f.get("_t")
    if t is None:
        t = f["_t"] = _transposed(f)
    return t


def _transposed(f):
    return {"wq_t": f["wq"].t().contiguous(), "wkv_t": f["wkv"].t().contiguous(), "wo_t": f["wo"].t().contiguous(),
            "w1_t": f["w1"].t().contiguous(), "w2_t": f["w2"].t().contiguous(),
            # [in][mirrored tap][out]: wconv_t[ci][t][co] = wconv[co][26 - t][ci] = cnn.weight[co, ci, 26 - t]
            "wconv_t": f["wconv"].view(256, 27, 256).flip(1).permute(2, 1, 0).reshape(256, 27 * 256).contiguous()}


def _fill(struct, tensors, names):
    for n in names:
        setattr(struct, n, tensors[n].data_ptr())
    return struct


def _forward_inference(image_feats, eps_block, eps_final, R, out_dim, pos_embed, norm_w, norm_b, deconv_w, deconv_b, *layer_params):
    """The forward alone (torch.no_grad(): evaluation, network.py:455-532 under `model.eval()`), every block in place:
    (image_feats [B, V = 4, C, D, H, W], ..., 15 tensors per layer ...) -> [B, 2R, 2R, 2R, out_dim].  (Round 2's single autograd
    node for the whole transformer lived here; its backward is the per-block nodes below since round 3 and was removed in
    round 5.)"""
    if not image_feats.is_cuda:
        raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
    lib = _lib()
    dev = image_feats.device
    n_layers = len(layer_params) // _NLP
    B, V, cond_dim = image_feats.shape[:3]
    S = image_feats.shape[3] * image_feats.shape[4] * image_feats.shape[5]
    M = B * R ** 3
    # `b v c d h w -> (b d h w) v c` (network.py:145-150) and the bf16 cast in one pass: per scene, the
    # [v c, d h w] matrix transposed
    feats = image_feats.detach().float().contiguous()
    cond_bf = torch.empty(B * S, V, cond_dim, dtype=torch.bfloat16, device=dev)
    with torch.cuda.device(dev):
        _check(lib.lara_batched_transpose(B, V * cond_dim, S, feats.data_ptr(), cond_bf.data_ptr(), 1, _stream(dev)),
               "lara_batched_transpose")
    x = volume_to_tokens(pos_embed.detach().float()).repeat(B, 1)       # network.py:152
    ws = _workspace(dev, "fwd", lib.lara_groupblock_workspace_bytes(B, R))
    with torch.cuda.device(dev):
        for l in range(n_layers):
            f = _layer_bf16([t.detach() for t in layer_params[l * _NLP:(l + 1) * _NLP]])
            w = _fill(_BlockWeights(), f, _GRAD_FIELDS)
            w.eps = eps_block
            _check(lib.lara_groupblock_forward(B, R, cond_dim, x.data_ptr(), cond_bf.data_ptr(), ctypes.byref(w),
                                               ws.data_ptr(), _stream(dev)), "lara_groupblock_forward")
        wd = deconv_w.detach().permute(2, 3, 4, 1, 0).reshape(8 * out_dim, 256).to(torch.bfloat16).contiguous()
        nw, nb, db = norm_w.detach().float().contiguous(), norm_b.detach().float().contiguous(), deconv_b.detach().float().contiguous()
        out = torch.empty(B, 2 * R, 2 * R, 2 * R, out_dim, dtype=torch.float32, device=dev)
        hws = _workspace(dev, "head", M * 512)
        _check(lib.lara_voltrans_head_forward(B, R, x.data_ptr(), nw.data_ptr(), nb.data_ptr(), float(eps_final),
                                              wd.data_ptr(), db.data_ptr(), out_dim, out.data_ptr(), hws.data_ptr(),
                                              _stream(dev)), "lara_voltrans_head_forward")
    return out


# ---------------------------------------------------------------------------------------------------------------
# The same forward / backward as ONE AUTOGRAD NODE PER BLOCK (the default): a block's 15 parameter gradients are
# released the moment its backward returns, so torch DDP's reducer (train_lightning.py:68-81) fills and all-reduces
# its buckets while the earlier blocks' backward is still running -- with one node for the whole transformer every gradient appears
# only when the whole 12-layer sweep is over and no bucket can overlap it.  Same kernels, same order, same bits.
#
#   image_feats --_CondFn--> (cond_bf16, token) ;  (pos_embed, token) --_StartFn--> x0 --_BlockFn x L--> x_L --_HeadFn--> out
#
# `token` is a one-element tensor whose only job is graph order: _StartFn's backward (after block 0's) hands it a
# gradient, which makes _CondFn's backward the LAST node to run -- it forms dL/d(image_feats) from the dK|dV every
# block left in the sweep's shared buffer with ONE product (K = layers * 512).
# ---------------------------------------------------------------------------------------------------------------
_ws_owner = {}     # device index -> (id of the sweep, layer) whose block backward used the "block_bwd" workspace last
_block_bwd_log = None   # tests: set to a list to record (event, layer) as the sweep runs


class _Sweep:
    """What the nodes of one forward share (not graph tensors)."""

    def __init__(self, B, R, cond_dim, n_layers, eps_block, dev):
        self.B, self.R, self.cond_dim, self.n_layers, self.eps_block, self.dev = B, R, cond_dim, n_layers, eps_block, dev
        self.cond_bf = None
        self.fs = [None] * n_layers       # bf16 operands of every layer (kept for the dcond product)
        self.dkv_all = None               # [rows, layers * 512] bf16: dK|dV of every block
        self.flat = None                  # all layers' gradient accumulators, one zero fill
        self.offs = self.shapes = self.per_layer = None

    def accumulators(self, l):
        f = self.fs[l]
        if self.flat is None:
            self.shapes = {n: tuple(f[n].shape) for n in _GRAD_FIELDS}
            self.offs, o = {}, 0
            for n in _GRAD_FIELDS:
                self.offs[n] = o
                o += (math.prod(self.shapes[n]) + 63) // 64 * 64
            self.per_layer = o
            self.flat = torch.zeros(self.n_layers * o, dtype=torch.float32, device=self.dev)
        base = l * self.per_layer
        return {n: self.flat[base + self.offs[n]:base + self.offs[n] + math.prod(self.shapes[n])].view(self.shapes[n])
                for n in _GRAD_FIELDS}


class _CondFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image_feats, sweep):
        lib = _lib()
        dev = image_feats.device
        B, V, C = image_feats.shape[:3]
        S = image_feats.shape[3] * image_feats.shape[4] * image_feats.shape[5]
        feats = image_feats.detach().float().contiguous()
        cond_bf = torch.empty(B * S, V, C, dtype=torch.bfloat16, device=dev)
        with torch.cuda.device(dev):
            _check(lib.lara_batched_transpose(B, V * C, S, feats.data_ptr(), cond_bf.data_ptr(), 1, _stream(dev)),
                   "lara_batched_transpose")
        sweep.cond_bf = cond_bf
        ctx.sweep, ctx.feat_shape, ctx.feat_dtype = sweep, tuple(image_feats.shape), image_feats.dtype
        return torch.zeros(1, dtype=torch.float32, device=dev)

    @staticmethod
    def backward(ctx, g_token):
        lib, sw = _lib(), ctx.sweep
        if not ctx.needs_input_grad[0]:
            return None, None
        dev = sw.dev
        cond_bf = sw.cond_bf
        dcond = torch.empty(cond_bf.shape, dtype=torch.float32, device=dev)
        lddkv = sw.n_layers * 512
        with torch.cuda.device(dev):
            wkv_all_t = torch.cat([f["wkv"] for f in sw.fs], 0).t().contiguous()
            _check(lib.lara_gemm_nt_bf16(sw.dkv_all.shape[0], sw.cond_dim, lddkv, sw.dkv_all.data_ptr(), wkv_all_t.data_ptr(),
                                         dcond.data_ptr(), 1, _stream(dev)), "lara_gemm_nt_bf16")
            Bf, V, C = ctx.feat_shape[:3]
            S = dcond.shape[0] // Bf
            d_feats = torch.empty(ctx.feat_shape, dtype=torch.float32, device=dev)
            _check(lib.lara_batched_transpose(Bf, S, V * C, dcond.data_ptr(), d_feats.data_ptr(), 0, _stream(dev)),
                   "lara_batched_transpose")
        sw.dkv_all = sw.flat = None
        return (d_feats if ctx.feat_dtype == torch.float32 else d_feats.to(ctx.feat_dtype)), None


class _StartFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos_embed, token, B, R):
        ctx.B, ctx.R = B, R
        return volume_to_tokens(pos_embed.detach().float()).repeat(B, 1)        # network.py:152

    @staticmethod
    def backward(ctx, g):
        # the same positional rows enter every scene
        return tokens_to_volume(g.view(ctx.B, ctx.R ** 3, 256).sum(0), 1, ctx.R), torch.zeros(1, dtype=torch.float32, device=g.device), None, None


class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sweep, l, *p):
        lib, sw = _lib(), sweep
        dev = x.device
        f = sw.fs[l] = _layer_bf16([t.detach() for t in p])
        w = _fill(_BlockWeights(), f, _GRAD_FIELDS)
        w.eps = sw.eps_block
        keep = _keep_activations()
        with torch.cuda.device(dev):
            if keep:
                nsave = lib.lara_groupblock_save_bytes(sw.B, sw.R)
                if nsave < 0:
                    _check(int(nsave), "lara_groupblock_save_bytes")
                act = torch.empty(nsave, dtype=torch.uint8, device=dev)
                x_out = torch.empty_like(x)
                _check(lib.lara_groupblock_forward_train(sw.B, sw.R, sw.cond_dim, x.data_ptr(), x_out.data_ptr(), sw.cond_bf.data_ptr(),
                                                         ctypes.byref(w), act.data_ptr(), _stream(dev)),
                       "lara_groupblock_forward_train")
            else:
                act = None
                x_out = x.clone()
                ws = _workspace(dev, "fwd", lib.lara_groupblock_workspace_bytes(sw.B, sw.R))
                _check(lib.lara_groupblock_forward(sw.B, sw.R, sw.cond_dim, x_out.data_ptr(), sw.cond_bf.data_ptr(), ctypes.byref(w),
                                                   ws.data_ptr(), _stream(dev)), "lara_groupblock_forward")
        ctx.save_for_backward(x)
        ctx.sweep, ctx.l, ctx.act = sw, l, act
        return x_out

    @staticmethod
    def backward(ctx, g):
        lib, sw, l = _lib(), ctx.sweep, ctx.l
        (x_in,) = ctx.saved_tensors
        dev = sw.dev
        f = sw.fs[l]
        ft = _layer_bf16_t(f)
        w = _fill(_BlockWeights(), f, _GRAD_FIELDS)
        w.eps = sw.eps_block
        wt = _fill(_BlockWeightsT(), ft, ("wq_t", "wkv_t", "wo_t", "w1_t", "w2_t", "wconv_t"))
        lddkv = sw.n_layers * 512
        if sw.dkv_all is None:
            sw.dkv_all = torch.empty(sw.cond_bf.shape[0] * sw.cond_bf.shape[1], lddkv, dtype=torch.bfloat16, device=dev)
        gd = sw.accumulators(l)
        dw = _fill(_BlockGrads(), gd, _GRAD_FIELDS)
        # `g` is overwritten in place with the gradient for the block's input and handed on.  That is safe because every
        # edge this node sits on is INTERNAL to `_forward_per_block` (the token tensors between `_StartFn`, the blocks and
        # `_HeadFn` never reach the caller, so no hook or `retain_grad` can hold one), and both producers of `g` -- the next
        # block and `_HeadFn.backward` -- hand over a buffer nobody else reads.
        g = g.float()
        if not g.is_contiguous():
            g = g.contiguous()
        # `chained`: the workspace still holds what block l + 1 of THIS sweep left (the bf16 copy of g, the neighbour table)
        chained = int(_ws_owner.get(dev.index) == (id(sw), l + 1))
        with torch.cuda.device(dev):
            ws = _workspace(dev, "block_bwd", lib.lara_groupblock_backward_workspace_bytes(sw.B, sw.R))
            if _block_bwd_log is not None:
                _block_bwd_log.append(("block_backward_launch", l))
            _check(lib.lara_groupblock_backward(sw.B, sw.R, sw.cond_dim, x_in.data_ptr(), sw.cond_bf.data_ptr(), ctypes.byref(w),
                                                ctypes.byref(wt), None if ctx.act is None else ctx.act.data_ptr(), g.data_ptr(), None,
                                                ctypes.byref(dw), chained, sw.dkv_all.data_ptr() + l * 1024, lddkv, ws.data_ptr(),
                                                _stream(dev)), "lara_groupblock_backward")
        _ws_owner[dev.index] = (id(sw), l)
        ctx.act = None      # released now, not with the graph; a second backward over a retained graph recomputes them (act = NULL)
        return (g, None, None, gd["ln1_w"], gd["ln1_b"], gd["wq"], gd["wkv"][:256], gd["wkv"][256:], gd["wo"], gd["ln2_w"], gd["ln2_b"],
                gd["w1"], gd["b1"], gd["w2"], gd["b2"], gd["ln3_w"], gd["ln3_b"],
                gd["wconv"].view(256, 3, 3, 3, 256).permute(0, 4, 1, 2, 3))


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, norm_w, norm_b, deconv_w, deconv_b, sweep, eps_final, out_dim):
        lib, sw = _lib(), sweep
        dev = x.device
        M = sw.B * sw.R ** 3
        with torch.cuda.device(dev):
            wd = deconv_w.detach().permute(2, 3, 4, 1, 0).reshape(8 * out_dim, 256).to(torch.bfloat16).contiguous()
            nw, nb, db = norm_w.detach().float().contiguous(), norm_b.detach().float().contiguous(), deconv_b.detach().float().contiguous()
            out = torch.empty(sw.B, 2 * sw.R, 2 * sw.R, 2 * sw.R, out_dim, dtype=torch.float32, device=dev)
            hws = _workspace(dev, "head", M * 512)
            _check(lib.lara_voltrans_head_forward(sw.B, sw.R, x.data_ptr(), nw.data_ptr(), nb.data_ptr(), float(eps_final),
                                                  wd.data_ptr(), db.data_ptr(), out_dim, out.data_ptr(), hws.data_ptr(),
                                                  _stream(dev)), "lara_voltrans_head_forward")
        ctx.save_for_backward(x, norm_w, norm_b, deconv_w)
        ctx.sweep, ctx.eps_final, ctx.out_dim = sw, float(eps_final), out_dim
        return out

    @staticmethod
    def backward(ctx, dout):
        lib, sw = _lib(), ctx.sweep
        x_last, norm_w, norm_b, deconv_w = ctx.saved_tensors
        dev, out_dim = sw.dev, ctx.out_dim
        M = sw.B * sw.R ** 3
        f32 = dict(dtype=torch.float32, device=dev)
        dout = dout.float().contiguous()
        g = torch.empty(M, 256, **f32)
        d_nw, d_nb = torch.zeros(256, **f32), torch.zeros(256, **f32)
        d_wd, d_b8 = torch.zeros(8 * out_dim, 256, **f32), torch.zeros(8 * out_dim, **f32)
        with torch.cuda.device(dev):
            wd_t = deconv_w.detach().permute(2, 3, 4, 1, 0).reshape(8 * out_dim, 256).to(torch.bfloat16).t().contiguous()
            nw, nb = norm_w.detach().float().contiguous(), norm_b.detach().float().contiguous()
            hws = _workspace(dev, "head_bwd", lib.lara_voltrans_head_backward_workspace_bytes(sw.B, sw.R, out_dim))
            _check(lib.lara_voltrans_head_backward(sw.B, sw.R, x_last.data_ptr(), nw.data_ptr(), nb.data_ptr(), ctx.eps_final,
                                                   wd_t.data_ptr(), out_dim, dout.data_ptr(), g.data_ptr(), d_nw.data_ptr(),
                                                   d_nb.data_ptr(), d_wd.data_ptr(), d_b8.data_ptr(), hws.data_ptr(),
                                                   _stream(dev)), "lara_voltrans_head_backward")
        _ws_owner.pop(dev.index, None)      # a new sweep starts: block L - 1 is not chained to anything
        # ... and with fresh gradient accumulators: `_CondFn.backward` releases them at the END of a sweep only when the image
        # features want a gradient; a second backward over a retained graph must not add into the first one's sums
        sw.flat = sw.dkv_all = None
        return (g, d_nw, d_nb, d_wd.view(2, 2, 2, out_dim, 256).permute(4, 3, 0, 1, 2), d_b8.view(8, out_dim).sum(0), None, None, None)


def _forward_per_block(module, image_feats):
    B, V, C = image_feats.shape[:3]
    R, n_layers = module.vol_low_res, len(module.layers)
    sweep = _Sweep(B, R, C, n_layers, float(module.layers[0].norm1.eps), image_feats.device)
    token = _CondFn.apply(image_feats, sweep)
    x = _StartFn.apply(module.pos_embed, token, B, R)
    for l, layer in enumerate(module.layers):
        x = _BlockFn.apply(x, sweep, l, *layer.flat_params())
    return _HeadFn.apply(x, module.norm.weight, module.norm.bias, module.deconv.weight, module.deconv.bias, sweep,
                         float(module.norm.eps), module.out_dim)


class GroupAttBlock(nn.Module):
    """Parameter container with the reference's attribute names (network.py:57-79)."""

    def __init__(self, inner_dim: int, cond_dim: int, num_heads: int, eps: float = 1e-5, attn_drop: float = 0.,
                 attn_bias: bool = False, mlp_ratio: float = 2., mlp_drop: float = 0.):
        super().__init__()
        if inner_dim != 256 or num_heads != 16 or attn_bias or mlp_ratio != 2. or attn_drop or mlp_drop:
            raise ValueError("kernels are specialised for LaRa's 256-dim, 16-head, bias-free, dropout-free blocks "
                             "(configs/base.yaml:17-20)")
        if cond_dim == inner_dim:
            # nn.MultiheadAttention then stores ONE fused in_proj_weight instead of q/k/v_proj_weight, a
            # different state_dict layout from the one the kernels' parameter list names
            raise ValueError("cond_dim must differ from inner_dim (LaRa: 800 vs 256); the fused in_proj_weight "
                             "layout of nn.MultiheadAttention is not supported")
        self.norm1 = nn.LayerNorm(inner_dim)
        self.cross_attn = nn.MultiheadAttention(embed_dim=inner_dim, num_heads=num_heads, kdim=cond_dim, vdim=cond_dim,
                                                dropout=attn_drop, bias=attn_bias, batch_first=True)
        self.cnn = nn.Conv3d(inner_dim, inner_dim, kernel_size=3, padding=1, bias=False)
        self.norm2 = nn.LayerNorm(inner_dim)
        self.norm3 = nn.LayerNorm(inner_dim)
        self.mlp = nn.Sequential(nn.Linear(inner_dim, int(inner_dim * mlp_ratio)), nn.GELU(), nn.Dropout(mlp_drop),
                                 nn.Linear(int(inner_dim * mlp_ratio), inner_dim), nn.Dropout(mlp_drop))

    def flat_params(self):
        sd = dict(self.named_parameters())
        return [sd[n] for n in _LAYER_PARAMS]


class VolTransformer(nn.Module):
    """Trainable drop-in for the reference ``VolTransformer`` (network.py:105-164): same constructor, same
    ``state_dict``, ``forward(image_feats [B, n_views, C, D, H, W]) -> [B, 2R, 2R, 2R, out_dim]``."""

    def __init__(self, embed_dim: int, image_feat_dim: int, n_groups: list, vol_low_res: int, vol_high_res: int,
                 out_dim: int, num_layers: int, num_heads: int, eps: float = 1e-6):
        super().__init__()
        if len(n_groups) != 1 or vol_low_res != 2 * n_groups[0] or vol_high_res != 2 * vol_low_res or out_dim % 16:
            raise ValueError("kernels are specialised for one group size with block_size 2 and a x2 deconvolution "
                             "(configs/base.yaml: n_groups [16], vol 32 -> 64)")
        self.vol_low_res, self.vol_high_res, self.out_dim, self.n_groups = vol_low_res, vol_high_res, out_dim, list(n_groups)
        self.embed_dim = embed_dim
        self.pos_embed = nn.Parameter(torch.randn(1, embed_dim, vol_low_res, vol_low_res, vol_low_res) * (1. / embed_dim) ** 0.5)
        self.layers = nn.ModuleList([GroupAttBlock(inner_dim=embed_dim, cond_dim=image_feat_dim, num_heads=num_heads, eps=eps)
                                     for _ in range(num_layers)])
        self.norm = nn.LayerNorm(embed_dim, eps=eps)
        self.deconv = nn.ConvTranspose3d(embed_dim, out_dim, kernel_size=2, stride=2, padding=0)

    def forward(self, image_feats: torch.Tensor) -> torch.Tensor:
        if not image_feats.is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        B, V, C, D = image_feats.shape[:4]
        if D != self.n_groups[0] or V != 4:
            raise RuntimeError("kernels are specialised for one image-feature voxel per group and 4 input views")
        if torch.is_grad_enabled():
            return _forward_per_block(self, image_feats)
        flat = [p for layer in self.layers for p in layer.flat_params()]
        return _forward_inference(image_feats, float(self.layers[0].norm1.eps), float(self.norm.eps), self.vol_low_res, self.out_dim,
                                 self.pos_embed, self.norm.weight, self.norm.bias, self.deconv.weight, self.deconv.bias, *flat)
