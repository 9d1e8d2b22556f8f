"""Fine-stage decoder (Decoder.forward_fine): the bench leg, then a breakdown of the fused path's wall time.  GPU box."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
if "--leg" in sys.argv:
    print(json.dumps(bench.fine_decoder_leg(dev), indent=1))
from torch.profiler import profile, ProfilerActivity
from oracle.finedec_ref import FineDecoderRef   # (module declarations only; nothing of the oracle is timed)
from lara_amd.fine import forward_fine
dec = FineDecoderRef().to(dev)
n = 524288
vol0, pf0, gout = torch.randn(n, 80, device=dev), torch.randn(4, 8, n, device=dev), torch.randn(n, 1, 12, device=dev)
def one():
    vol, pfp = vol0.clone().requires_grad_(True), pf0.clone().requires_grad_(True)
    forward_fine(dec, vol, torch.einsum('lcb->blc', pfp)).backward(gout)
for _ in range(2):
    one()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    one()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=70))
