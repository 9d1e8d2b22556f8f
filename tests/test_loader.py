"""`lara_amd.dataset`: the adapter around the REFERENCE's own dataset class (dataLoader/gobjverse.py:17-146).  The
class itself is not restated anywhere in this repository; tests/golden/loader_ref.npz holds items it returned on a
small in-memory scene store (tests/golden/make_loader_fixture.py).  Checked here: `skip_cpu_rays` removes the CPU
`build_rays` work from the reference class without changing any other field (needs /root/reference: build container
only), and `collate_to_device` rebuilds both ray maps on the GPU equal to the reference's CPU rays."""
import os
import random
import sys
import types

import numpy as np
import pytest
import torch

from tests.helpers_store import make_store

FX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loader_ref.npz")
REF = "/root/reference"


def _fixture_item(fx, tag, index):
    """One item of the reference class, as stored in the fixture."""
    pre = f"{tag}.{index}."
    it = {k[len(pre):]: fx[k] for k in fx.files if k.startswith(pre)}
    scene, view = str(it.pop("scene")), it.pop("tar_view").tolist()
    it["meta"] = {"scene": scene, "tar_view": view, "frame_id": 0, "tar_h": 32, "tar_w": 32}
    it["fovx"], it["fovy"] = it["fovx"][()], it["fovy"][()]
    return it


def test_skip_cpu_rays_swaps_and_returns_the_original():
    from lara_amd import dataset
    mod = types.ModuleType("fake_loader")
    mod.build_rays = lambda *a, **k: "cpu rays"
    orig = dataset.skip_cpu_rays(mod)
    assert orig() == "cpu rays"
    out = mod.build_rays(np.eye(4)[None], np.eye(3)[None], 32, 32, 1.0 / 16)
    assert isinstance(out, np.ndarray) and out.size == 0


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "dataLoader")), reason="needs the reference checkout (build container only)")
def test_reference_class_with_the_hook_returns_the_same_items_minus_the_rays():
    from lara_amd import dataset
    store = make_store(seed=5)
    saved = {k: sys.modules.get(k) for k in ("h5py", "cv2", "dataLoader", "dataLoader.gobjverse", "dataLoader.utils")}
    h5 = types.ModuleType("h5py")
    h5.File = lambda path, mode="r": store
    pkg = types.ModuleType("dataLoader")       # bare package: only the loader module itself is executed
    pkg.__path__ = [os.path.join(REF, "dataLoader")]
    sys.modules.update({"h5py": h5, "dataLoader": pkg})
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.modules.pop("dataLoader.gobjverse", None)
    sys.path.insert(0, REF)
    try:
        import dataLoader.gobjverse as ref
        orig = dataset.skip_cpu_rays(ref)
        fx = np.load(FX)
        cfg = types.SimpleNamespace(data_root="mem", split="test", img_size=(32, 32), n_group=4, n_scenes=100, load_normal=True)
        ds = ref.gobjverse(cfg)
        for index in (0, len(ds) - 1):
            random.seed(100 + index)
            torch.manual_seed(200 + index)
            it = ds[index]
            assert it["tar_rays"].size == 0 and it["tar_rays_down"].size == 0
            want = _fixture_item(fx, "test4", index)
            for k, v in want.items():
                if k in dataset.RAY_KEYS or k == "meta":
                    continue
                np.testing.assert_array_equal(np.asarray(it[k]), v, err_msg=k)
        ref.build_rays = orig
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@pytest.mark.gpu
def test_collate_builds_the_reference_rays_on_the_device():
    from lara_amd.dataset import collate_to_device
    fx = np.load(FX)
    idx = (0, int(fx["test4.len"]) - 1)
    items = [_fixture_item(fx, "test4", i) for i in idx]
    stubbed = [dict(it, tar_rays=np.zeros((0,), np.float32), tar_rays_down=np.zeros((0,), np.float32)) for it in items]
    for its in (items, stubbed):        # items carrying the reference's CPU rays, and items from a hooked loader
        batch = collate_to_device(its)
        assert batch["tar_rays"].shape == (2, 8, 32, 32, 6) and batch["tar_rays_down"].shape == (2, 8, 2, 2, 6)
        assert batch["tar_rgb"].is_cuda and batch["tar_msk"].dtype == torch.uint8
        for b, i in enumerate(idx):
            for k in ("tar_rays", "tar_rays_down"):
                want = fx[f"test4.{i}.{k}"]
                got = batch[k][b].cpu().numpy()
                assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), k


def test_worker_side_collate_is_picklable_and_runs_in_dataloader_workers():
    """`collate_cpu` (the half that may run inside DataLoader workers) is a module-level function, drops the two ray keys and
    needs no GPU: it is driven here by a real DataLoader with two worker processes."""
    import pickle
    from lara_amd import dataset
    pickle.dumps(dataset.collate_cpu)
    fx = np.load(FX)
    items = [_fixture_item(fx, "test4", i) for i in (0, int(fx["test4.len"]) - 1)]

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 4

        def __getitem__(self, i):
            return dict(items[i % 2], tar_rays=np.zeros((0,), np.float32), tar_rays_down=np.zeros((0,), np.float32))
    loader = torch.utils.data.DataLoader(DS(), batch_size=2, num_workers=2, collate_fn=dataset.collate_cpu)
    batches = list(loader)
    assert len(batches) == 2
    for b in batches:
        assert "tar_rays" not in b and "tar_rays_down" not in b
        assert b["tar_c2w"].shape == (2, 8, 4, 4) and not b["tar_c2w"].is_cuda


@pytest.mark.gpu
def test_on_device_finishes_cpu_collated_batches():
    from lara_amd import dataset
    fx = np.load(FX)
    idx = (0, int(fx["test4.len"]) - 1)
    items = [_fixture_item(fx, "test4", i) for i in idx]
    got = list(dataset.on_device([dataset.collate_cpu(items)], "cuda"))[0]
    want = dataset.collate_to_device(items)
    for k in ("tar_rays", "tar_rays_down", "tar_c2w", "tar_rgb"):
        assert got[k].is_cuda and torch.equal(got[k], want[k]), k
