"""Generates tests/golden/loader_ref.npz by running the REFERENCE's own dataset class
(/root/reference/dataLoader/gobjverse.py:17-146) on a small in-memory scene store (tests/helpers_store.py) through a
stand-in `h5py` module -- h5py is not installed in the build image; the class only uses `h5py.File(path, 'r')` and the
group interface.  Run in the build container only:  python tests/golden/make_loader_fixture.py"""
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers_store import make_store  # noqa: E402

store = make_store(seed=5)
h5 = types.ModuleType("h5py")
h5.File = lambda path, mode="r": store
sys.modules["h5py"] = h5
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
# the package's __init__ imports every dataset (imageio, PIL, ...): register a bare package so that only the loader
# module itself (and dataLoader.utils) is executed
pkg = types.ModuleType("dataLoader")
pkg.__path__ = ["/root/reference/dataLoader"]
sys.modules["dataLoader"] = pkg
sys.path.insert(0, "/root/reference")
from dataLoader.gobjverse import gobjverse  # noqa: E402

out = {}
for split, n_group, tag in (("train", 4, "train4"), ("test", 4, "test4"), ("test", 1, "test1")):
    cfg = types.SimpleNamespace(data_root="mem", split=split, img_size=(32, 32), n_group=n_group, n_scenes=100, load_normal=True)
    ds = gobjverse(cfg)
    out[f"{tag}.len"] = np.array(len(ds))
    for index in (0, len(ds) - 1):
        random.seed(100 + index)
        torch.manual_seed(200 + index)
        it = ds[index]
        for k, v in it.items():
            if k == "meta":
                out[f"{tag}.{index}.scene"] = np.array(str(v["scene"]))
                out[f"{tag}.{index}.tar_view"] = np.array(v["tar_view"])
            else:
                out[f"{tag}.{index}.{k}"] = np.asarray(v)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "loader_ref.npz"), **out)
print("wrote", len(out), "arrays")
