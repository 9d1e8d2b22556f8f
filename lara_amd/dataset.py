"""The gobjaverse HDF5 scene store -> LaRa's batch dictionary (the loader half).

Mirror of the reference's ``dataLoader/gobjverse.py:17-146`` dataset class: the same HDF5 layout (one group per
scene holding ``image_<i>`` RGBA uint8 [H,W,4], ``normal_<i>`` uint8 [H,W,3], ``c2w_<i>`` [4,4], ``fov_<i>`` [2] and
``groups/groups_<n>_<k>`` view-id lists; optional top-level ``splits``), the same view selection (same calls on
`random` / `torch.randperm` in the same order, so a seeded run picks the same views), the same background
compositing, first-view alignment and near/far -- and the same keys, shapes and dtypes in the returned dictionary.
What differs, by design: ``tar_rays`` / ``tar_rays_down`` (6 floats per pixel and view, 50 MB per scene at 8 x 512^2)
are NOT built on the CPU workers and shipped through the DataLoader; ``collate_to_device`` moves the cameras and
images to the GPU and generates both ray maps there with the HIP kernel (``lara_amd.batch.build_rays``).

``store`` is anything with the h5py group interface (``keys()``, ``[name]``, array-like leaves); ``open_hdf5(path)``
opens a real file with h5py, which is imported lazily (it is not installed in the build image: the tests run the
class against an in-memory store, and against outputs of the reference's own class on that store)."""
from __future__ import annotations

import random

import numpy as np
import torch

B2C = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], dtype=np.float32)   # gobjverse.py:37


def open_hdf5(path):
    try:
        import h5py
    except ImportError as e:   # pragma: no cover
        raise RuntimeError("lara_amd.dataset.open_hdf5 needs h5py (the reference's loader does too: dataLoader/gobjverse.py:8)") from e
    return h5py.File(path, "r")


def fov_to_ixt(fov, reso):                      # gobjverse.py:10-15
    ixt = np.eye(3, dtype=np.float32)
    ixt[0][2], ixt[1][2] = reso[0] / 2, reso[1] / 2
    focal = .5 * reso / np.tan(.5 * fov)
    ixt[[0, 1], [0, 1]] = focal
    return ixt


class GobjverseScenes(torch.utils.data.Dataset):
    def __init__(self, store, split="train", img_size=(512, 512), n_group=4, n_scenes=1 << 30, load_normal=False):
        self.store, self.split = store, split
        self.img_size = np.array(img_size)
        self.n_group, self.load_normal = n_group, load_normal
        scenes_name = np.array(sorted(store.keys()))
        if "splits" in scenes_name:             # gobjverse.py:28-29 (the reference reads the test list for every split)
            self.scenes_name = np.asarray(store["splits"]["test"][:]).astype(str)
        else:                                   # gobjverse.py:31-35: every 10th scene is a test scene
            i_test = np.arange(len(scenes_name))[::10][:n_scenes]
            i_train = np.array([i for i in np.arange(len(scenes_name)) if (i not in i_test)])[:n_scenes]
            self.scenes_name = scenes_name[i_train] if split == "train" else scenes_name[i_test]

    def __len__(self):
        return len(self.scenes_name)

    def _views(self, scene):                    # gobjverse.py:46-54
        g, n = scene["groups"], self.n_group
        if self.split == "train" and n > 1:
            src = [random.choices(g[f"groups_{n}_{i}"])[0] for i in torch.randperm(n).tolist()]
            return src, src + [random.choices(g[f"groups_{n}_{i}"])[0] for i in torch.randperm(n).tolist()]
        if n == 1:
            src = [g["groups_4_0"][0]]
        else:
            src = [g[f"groups_{n}_{i}"][0] for i in range(n)]
        return src, src + [g[f"groups_4_{i}"][-1] for i in range(4)]

    def __getitem__(self, index):
        name = self.scenes_name[index]
        scene = self.store[name]
        _, view_id = self._views(scene)
        imgs, bgs, nrms, msks, c2ws, w2cs, ixts = [], [], [], [], [], [], []
        for i, idx in enumerate(view_id):       # gobjverse.py:98-122
            if self.split != "train" or i < self.n_group:
                bg = np.ones(3).astype(np.float32)
            else:
                bg = np.ones(3).astype(np.float32) * random.choice([0.0, 0.5, 1.0])
            img = np.array(scene[f"image_{idx}"])
            msks.append((img[..., -1] > 0).astype("uint8"))
            img = img.astype(np.float32) / 255.
            imgs.append((img[..., :3] * img[..., -1:] + bg * (1 - img[..., -1:])).astype(np.float32))
            if self.load_normal:
                nrms.append(np.array(scene[f"normal_{idx}"]).astype(np.float32) / 255. * 2 - 1.0)
            c2w = np.array(scene[f"c2w_{idx}"], dtype=np.float32)
            c2ws.append(c2w)
            w2cs.append(np.linalg.inv(c2w))
            ixts.append(fov_to_ixt(np.array(scene[f"fov_{idx}"], dtype=np.float32), self.img_size))
            bgs.append(bg)
        tar_c2ws, tar_w2cs = np.stack(c2ws), np.stack(w2cs)
        # align cameras using the first view (gobjverse.py:57-64)
        r = np.linalg.norm(tar_c2ws[0, :3, 3])
        ref_c2w = np.eye(4, dtype=np.float32).reshape(1, 4, 4)
        ref_w2c = np.eye(4, dtype=np.float32).reshape(1, 4, 4)
        ref_c2w[:, 2, 3], ref_w2c[:, 2, 3] = -r, r
        transform_mats = ref_c2w @ tar_w2cs[:1]
        tar_w2cs = tar_w2cs.copy() @ tar_c2ws[:1] @ ref_w2c
        tar_c2ws = transform_mats @ tar_c2ws.copy()
        H, W = self.img_size
        ret = {"fovx": scene["fov_0"][0], "fovy": scene["fov_0"][1],
               "tar_c2w": tar_c2ws, "tar_w2c": tar_w2cs, "tar_ixt": np.stack(ixts), "tar_rgb": np.stack(imgs),
               "tar_msk": np.stack(msks), "transform_mats": transform_mats, "bg_color": np.stack(bgs)}
        if self.load_normal:
            tar_nrms = np.stack(nrms) @ transform_mats[0, :3, :3].T
            ret["tar_nrm"] = tar_nrms.transpose(1, 0, 2, 3).reshape(H, len(view_id) * W, 3)
        ret["near_far"] = np.array([r - 0.8, r + 0.8]).astype(np.float32)
        ret["meta"] = {"scene": name, "tar_view": view_id, "frame_id": 0, "tar_h": int(H), "tar_w": int(W)}
        return ret


def collate_to_device(items, device="cuda"):
    """Default-collate a list of `GobjverseScenes` items onto `device` and add ``tar_rays`` [B,V,H,W,6] and
    ``tar_rays_down`` [B,V,H/16,W/16,6] there (the reference's loader computes them per scene on the CPU,
    gobjverse.py:90-93; dataLoader/utils.py:21-34)."""
    from .batch import build_rays
    dev = torch.device(device)
    batch = torch.utils.data.default_collate(items)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    B, V = batch["tar_c2w"].shape[:2]
    H, W = int(batch["meta"]["tar_h"][0]), int(batch["meta"]["tar_w"][0])
    c2w, ixt = batch["tar_c2w"].reshape(B * V, 4, 4).float(), batch["tar_ixt"].reshape(B * V, 3, 3).float()
    batch["tar_rays"] = build_rays(c2w, ixt, H, W, 1.0).view(B, V, H, W, 6)
    down = build_rays(c2w, ixt, H, W, 1.0 / 16)
    batch["tar_rays_down"] = down.view(B, V, *down.shape[1:])
    return batch
