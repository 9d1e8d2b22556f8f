"""Adapter between the reference's OWN gobjaverse dataset class and the device-side ray kernel (SURVEY.md
section 8f row 3, the loader half).

The dataset class stays reference code (``dataLoader/gobjverse.py:17-146``: HDF5 layout, view selection,
compositing, first-view alignment are its business and are not restated here).  What this module changes is only
where the two ray maps come from: the reference builds ``tar_rays`` / ``tar_rays_down`` (6 floats per pixel and
view, 50 MB per scene at 8 x 512^2) on the CPU workers (``gobjverse.py:90-93``, ``dataLoader/utils.py:21-34``) and
ships them through the DataLoader; here

* ``skip_cpu_rays(module)`` turns the two ``build_rays`` calls of the reference's loader module into no-ops, and
* ``collate_to_device(items, device)`` is the ``collate_fn`` that moves the collated cameras and images to the GPU
  and generates both ray maps there with the HIP kernel (``lara_amd.batch.build_rays``, csrc/rays.hip).

    import dataLoader.gobjverse as ref                       # the reference's module, unmodified
    lara_amd.dataset.skip_cpu_rays(ref)
    loader = DataLoader(ref.gobjverse(cfg), batch_size=4, num_workers=0, collate_fn=lambda b: collate_to_device(b, "cuda"))

Worker processes: a ``collate_fn`` runs INSIDE the DataLoader's workers, where HIP cannot be initialised after a fork (and a
lambda does not pickle under spawn) -- so the form above is for ``num_workers=0``.  With workers (the reference trains with 8,
train_lightning.py:35-39) keep the default CPU collate in the workers and finish the batch in the training process:

    lara_amd.dataset.skip_cpu_rays(ref)                      # before the workers start; patches this process AND, under fork,
                                                             # the workers (under spawn call it in `worker_init_fn` as well)
    loader = DataLoader(ref.gobjverse(cfg), batch_size=4, num_workers=8, collate_fn=lara_amd.dataset.collate_cpu)
    for batch in lara_amd.dataset.on_device(loader, "cuda"): ...
"""
from __future__ import annotations

import numpy as np
import torch

RAY_KEYS = ("tar_rays", "tar_rays_down")


def skip_cpu_rays(loader_module):
    """Replace ``build_rays`` in the namespace of the reference's loader module (``dataLoader.gobjverse``) by a stub
    returning an empty array: its ``__getitem__`` then spends no CPU time on the rays and ships 0 bytes for them;
    ``collate_to_device`` fills both keys on the GPU.  Returns the original function (to restore it)."""
    original = loader_module.build_rays
    loader_module.build_rays = lambda c2ws, ixts, H, W, scale=1.0: np.zeros((0,), dtype=np.float32)
    return original


def collate_cpu(items):
    """The worker-side half: default collate WITHOUT the two ray maps (picklable: a module-level function)."""
    return torch.utils.data.default_collate([{k: v for k, v in it.items() if k not in RAY_KEYS} for it in items])


def finish_on_device(batch, device="cuda"):
    """The training-process half: move a CPU-collated batch (``collate_cpu``) to `device` and build both ray maps there."""
    from .batch import build_rays
    dev = torch.device(device)
    batch = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
    B, V = batch["tar_c2w"].shape[:2]
    H, W = int(batch["meta"]["tar_h"][0]), int(batch["meta"]["tar_w"][0])
    c2w, ixt = batch["tar_c2w"].reshape(B * V, 4, 4).float(), batch["tar_ixt"].reshape(B * V, 3, 3).float()
    batch["tar_rays"] = build_rays(c2w, ixt, H, W, 1.0).view(B, V, H, W, 6)
    down = build_rays(c2w, ixt, H, W, 1.0 / 16)
    batch["tar_rays_down"] = down.view(B, V, *down.shape[1:])
    return batch


def on_device(loader, device="cuda"):
    """Iterate a DataLoader built with ``collate_fn=collate_cpu`` (any ``num_workers``), yielding device batches."""
    for batch in loader:
        yield finish_on_device(batch, device)


def collate_to_device(items, device="cuda"):
    """Default-collate a list of the reference dataset's items onto `device`; ``tar_rays`` [B,V,H,W,6] and
    ``tar_rays_down`` [B,V,H/16,W/16,6] are (re)built there from the collated ``tar_c2w`` / ``tar_ixt``, whatever
    the items carried under those keys (CPU rays, or the empty stubs of `skip_cpu_rays`).  ``num_workers=0`` only (see the
    module docstring): it touches the GPU."""
    return finish_on_device(collate_cpu(items), device)
