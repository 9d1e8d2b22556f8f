"""The bench's `fine_stage` leg alone (one scene through LaRa's render section after start_fine).  GPU box."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
a = argparse.Namespace(views=8, res=512, grid=64, regime=sys.argv[1] if len(sys.argv) > 1 else "init")
print(json.dumps(bench.fine_stage_leg(torch.device("cuda:0"), a), indent=1))
