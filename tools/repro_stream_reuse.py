#!/usr/bin/env python
"""Reproducer for DESIGN.md section 8.3 (the round-3 device hang / memory fault), reduced to its mechanism and made harmless:
every index stays in bounds, so the race shows up as WRONG VALUES instead of a dead device.

(1) `reuse_under_reader`: a tensor allocated on the caller's stream and read by a kernel queued on a side stream is freed on
    the host while that kernel is still waiting; the caching allocator hands the block back to the caller's stream AT ONCE and
    the next allocation there overwrites it under the reader.  `Tensor.record_stream(side)` (what `pipeline.hand_over` does)
    makes the free wait.  In the pipeline the overwritten tensor was the fine pass's subset INDICES (`masks[i].nonzero()`, made
    on the caller's stream, read by `index_copy_` in `_TakeRows.backward` on a scene stream): garbage indices -> writes
    anywhere -> memory fault, or a corrupted list length -> a kernel that never ends.
(2) `accumulation_allocates_on_consumer_stream`: why only the variant whose five gradients were VIEWS OF ONE BUFFER died.
    Autograd adds the gradients that meet in one input slot on the CONSUMER's stream (here the caller's: `UnbindBackward`).
    With a gradient that owns its storage it adds in place (`can_accumulate_inplace`: storage use count 1) -- no allocation.
    Views of a shared buffer fail that test, so every meeting point becomes `old + new` = a fresh allocation from the CALLER's
    pool in the MIDDLE of the backward, while the other scene stream (which the caller's stream has not waited for) still
    has index_copy_ kernels queued that read blocks the host already freed: exactly (1).
Run on a GPU box: python tools/repro_stream_reuse.py   (prints one JSON line; tests/test_stream_safety_gpu.py asserts on it)
"""
import json

import torch


def reuse_under_reader(protect: bool, n: int = 1 << 18) -> dict:
    dev = torch.device("cuda:0")
    cur, side = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    idx = torch.arange(n, device=dev)                  # 2 MB of int64 indices, from the caller's pool
    src = torch.ones(n, device=dev)
    ptr = idx.data_ptr()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        torch.cuda._sleep(400_000_000)                 # the side stream lags behind the host, as a scene stream does
        out = torch.zeros(n, device=dev).index_copy_(0, idx, src)
    if protect:
        idx.record_stream(side)
    del idx                                            # last reference gone: freed on the host's clock
    clobber = torch.zeros(n, dtype=torch.int64, device=dev)   # caller's stream, same size class; all-zero = in-bounds indices
    reused = clobber.data_ptr() == ptr
    torch.cuda.synchronize()
    return {"protect": protect, "block_reused_while_reader_pending": bool(reused),
            "rows_written": int((out == 1).sum()), "rows_expected": n}


class _TwoGrads(torch.autograd.Function):
    """x -> (x, x); backward returns either two own tensors or two views of one buffer."""
    @staticmethod
    def forward(ctx, a, b, carved):
        ctx.carved = carved
        return a * 1.0, b * 1.0

    @staticmethod
    def backward(ctx, ga, gb):
        if ctx.carved:
            buf = torch.cat([ga.reshape(-1), gb.reshape(-1)])
            n = ga.numel()
            return buf[:n].view_as(ga), buf[n:].view_as(gb), None
        return ga.clone(), gb.clone(), None


def _segment_stream(ptr: int):
    for seg in torch.cuda.memory_snapshot():
        if seg["address"] <= ptr < seg["address"] + seg["total_size"]:
            return seg["stream"]
    return None


def accumulation_allocates_on_consumer_stream(carved: bool, n: int = 1 << 20) -> dict:
    dev = torch.device("cuda:0")
    cur, side = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)
    x = torch.randn(2, n, device=dev, requires_grad=True)
    a, b = x.unbind(0)                                 # UnbindBackward lives on the caller's stream
    seen = {}
    a.register_hook(lambda g: seen.update(ptr=g.data_ptr()))
    side.wait_stream(cur)
    with torch.cuda.stream(side):                      # two consumers of `a` on the side stream: their gradients meet in a's slot
        p, q = _TwoGrads.apply(a, b, carved)           # (in the pipeline: the subset gather and the coarse views' rasteriser node)
        r, t = _TwoGrads.apply(a, b, carved)
        loss = p.sum() + q.sum() + 2.0 * (r.sum() + t.sum())
    cur.wait_stream(side)
    loss.backward()
    torch.cuda.synchronize()
    where = _segment_stream(seen["ptr"])
    return {"carved": carved, "accumulated_on_callers_pool": where == cur.cuda_stream, "segment_stream": where,
            "caller_stream": cur.cuda_stream, "side_stream": side.cuda_stream, "grad_ok": bool((x.grad[0] == 3).all())}


if __name__ == "__main__":
    print(json.dumps({"reuse": [reuse_under_reader(False), reuse_under_reader(True)],
                      "accumulate": [accumulation_allocates_on_consumer_stream(False), accumulation_allocates_on_consumer_stream(True)]}))
