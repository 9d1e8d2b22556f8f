"""Cross-stream lifetime of the tensors the two-stream pipeline passes between the caller's stream and the scene streams
(DESIGN.md section 8.3): the mechanism behind round 3's device hang, reduced (tools/repro_stream_reuse.py) and the guard
(`lara_amd.pipeline.hand_over` = `Tensor.record_stream`) that closes it."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
pytestmark = pytest.mark.gpu


def test_block_freed_on_the_callers_stream_is_overwritten_under_a_side_stream_reader_unless_recorded():
    import repro_stream_reuse as rr
    bad, good = rr.reuse_under_reader(False), rr.reuse_under_reader(True)
    # unprotected: the allocator reuses the block at once and the queued index_copy_ reads the new owner's zeros
    assert bad["block_reused_while_reader_pending"] and bad["rows_written"] < bad["rows_expected"], bad
    # record_stream: the free waits for the reader
    assert not good["block_reused_while_reader_pending"] and good["rows_written"] == good["rows_expected"], good


def test_gradient_views_of_one_buffer_turn_autograds_accumulation_into_allocations_on_the_consumers_stream():
    import repro_stream_reuse as rr
    own, carved = rr.accumulation_allocates_on_consumer_stream(False), rr.accumulation_allocates_on_consumer_stream(True)
    assert own["grad_ok"] and carved["grad_ok"]
    assert not own["accumulated_on_callers_pool"], own          # in place, in the producer's (side-stream) block
    assert carved["accumulated_on_callers_pool"], carved        # old + new: a fresh block from the caller's pool, mid-backward


def test_hand_over_walks_tensors_cameras_and_containers():
    from lara_amd import cameras
    from lara_amd.pipeline import hand_over
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    cams = cameras.make_cameras(cameras.turntable_c2w(2), 32, 32, 0.75, 0.75, 0.5, 2.5, device=dev)
    t = torch.zeros(1 << 18, device=dev)
    ptr = t.data_ptr()
    hand_over({"a": [t, None, 3], "c": cams, "cpu": torch.zeros(2)}, side)
    with torch.cuda.stream(side):
        torch.cuda._sleep(200_000_000)
    del t
    again = torch.zeros(1 << 18, device=dev)
    assert again.data_ptr() != ptr          # the recorded stream has pending work: the block is not handed out again yet
    torch.cuda.synchronize()
