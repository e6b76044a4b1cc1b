"""On-disk sample format of the reference and a synthetic generator for it.

A sample is an ``.npz`` with ``raw`` (H x W x C integer or float sensor values, C = 4
packed-Bayer planes or 3) and ``max_val`` (the white level).  A training item is the
normalised high-resolution image ``raw / max_val`` as 1 x C x H x W (odd trailing
row/column dropped) and its 2 x 2 average-pooled low-resolution input, with the
white level and file name alongside (data/loader.py:20-61, :100-124 of the reference;
utils/util.py:313-321 for the pooling).  Files are read with ``numpy.load`` in its
default non-pickle mode.  The reference's dataset itself is not shipped, so
``write_synthetic_npz`` makes files of the same shape for the harness and tests.
"""
from __future__ import annotations

import glob
import os
from typing import Dict, List

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .harness import downsample2


def to_bchw(img: torch.Tensor) -> torch.Tensor:
    """H x W x C or B x H x W x C (C in {3, 4}) -> B x C x H x W; B x C x H x W passes through."""
    if img.dim() == 3:
        return img.permute(2, 0, 1).unsqueeze(0)
    if img.dim() == 4 and img.shape[1] not in (3, 4) and img.shape[-1] in (3, 4):
        return img.permute(0, 3, 1, 2)
    return img


class RawPatchDataset(Dataset):
    """All ``*.npz`` under `data_dir`, sorted by name; items as described in the module docstring."""

    def __init__(self, data_dir: str):
        self.paths: List[str] = sorted(glob.glob(os.path.join(data_dir, "*.npz")))

    def __len__(self) -> int:
        return len(self.paths)

    def __getitem__(self, idx: int) -> Dict:
        path = self.paths[idx]
        with np.load(path) as f:
            raw, max_val = f["raw"], float(np.asarray(f["max_val"]).reshape(-1)[0])
        # float64 division and pooling, rounded to float32 once: what the reference's numpy / torch calls amount to
        # (float32 array / 0-d integer array promotes to float64; the training loop casts with .float())
        hr = to_bchw(torch.from_numpy(raw.astype(np.float32).astype(np.float64) / max_val))
        hr = hr[:, :, : hr.shape[2] & ~1, : hr.shape[3] & ~1].contiguous()
        return {"lr": downsample2(hr).float(), "hr": hr.float(), "max": max_val, "filename": os.path.basename(path)}


def collate(items: List[Dict]) -> Dict:
    """Stack items (each 1 x C x H x W, so batches are B x 1 x C x H x W as the reference's loaders give them)."""
    items = [it for it in items if it is not None]
    if not items:
        raise ValueError("empty batch")
    return {"lr": torch.stack([it["lr"] for it in items]), "hr": torch.stack([it["hr"] for it in items]),
            "max": torch.tensor([float(it["max"]) for it in items]), "filename": [it["filename"] for it in items]}


def loaders(train_dir: str, val_dir: str, batch_size: int, num_workers: int = 0):
    """Shuffled training loader at `batch_size`, validation loader at batch size 1 (data/loader.py:175-223)."""
    train = DataLoader(RawPatchDataset(train_dir), batch_size=batch_size, shuffle=True, num_workers=num_workers, collate_fn=collate)
    val = DataLoader(RawPatchDataset(val_dir), batch_size=1, shuffle=False, num_workers=num_workers, collate_fn=collate)
    return train, val


def prepare_batch_gpu(raw: torch.Tensor, max_val) -> Dict[str, torch.Tensor]:
    """The dataset item for a batch of sensor samples that already sit in GPU memory: `raw` is B x H x W x C
    (uint16 counts or float32, C <= 4) on a ROCm device, `max_val` a number or B white levels.  Returns
    {"hr": B x C x H2 x W2, "lr": B x C x H2/2 x W2/2} fp32, bit-identical to `RawPatchDataset` items of the same
    samples (rst_raw_prepare: normalisation and the 2 x 2 average pool in one kernel).  No CPU fallback."""
    from . import _lib
    from ._lib import check
    lib = _lib.load()
    if not raw.is_cuda:
        raise RuntimeError(f"prepare_batch_gpu: raw is on {raw.device}; this operator only runs on a ROCm GPU")
    if raw.dim() == 3:
        raw = raw.unsqueeze(0)
    if raw.dim() != 4 or raw.shape[-1] > 4 or raw.dtype not in (torch.uint16, torch.int16, torch.float32):
        raise RuntimeError(f"prepare_batch_gpu: expected B x H x W x C (C <= 4) uint16 or float32, got {raw.dtype} {tuple(raw.shape)}")
    raw = raw.contiguous()
    B, H, W, C = raw.shape
    mv = torch.as_tensor(max_val, dtype=torch.float32).reshape(-1).to(raw.device)
    if mv.numel() == 1:
        mv = mv.expand(B)
    if mv.numel() != B:
        raise RuntimeError(f"prepare_batch_gpu: {mv.numel()} white levels for {B} samples")
    mv = mv.contiguous()
    hr = torch.empty((B, C, H & ~1, W & ~1), dtype=torch.float32, device=raw.device)
    lr = torch.empty((B, C, (H & ~1) // 2, (W & ~1) // 2), dtype=torch.float32, device=raw.device)
    check(lib.rst_raw_prepare(raw.data_ptr(), int(raw.dtype != torch.float32), mv.data_ptr(), hr.data_ptr(), lr.data_ptr(),
                              B, H, W, C, torch.cuda.current_stream().cuda_stream), "rst_raw_prepare")
    return {"lr": lr, "hr": hr}


def write_synthetic_npz(out_dir: str, n: int, size: int = 64, channels: int = 4, max_val: int = 1023, seed: int = 0) -> List[str]:
    """`n` smooth random `size` x `size` x `channels` uint16 images with white level `max_val`."""
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    paths = []
    for i in range(n):
        coarse = rng.random((size // 8 + 1, size // 8 + 1, channels))
        img = np.kron(coarse, np.ones((8, 8, 1)))[:size, :size] * 0.9 + rng.random((size, size, channels)) * 0.1
        path = os.path.join(out_dir, f"{i:05d}.npz")
        np.savez(path, raw=np.round(img * max_val).astype(np.uint16), max_val=np.array(max_val))
        paths.append(path)
    return paths
