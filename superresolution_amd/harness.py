"""Training / validation harness around any x2 super-resolution ``nn.Module``
(the HIP-backed ``Restormer`` on a GPU, a stand-in on CPU for tests).

Behavioural counterparts of the reference's scripts, so that its checkpoints
and its numbers carry over:

* the training step order of model/train.py:82-90 -- zero grads, forward,
  loss(output, hr), backward, clip the global gradient norm to 1.0, Adam
  (lr 1e-4, weight decay 1e-6), CosineAnnealingLR(T_max=epochs, eta_min=1e-6)
  stepped once per epoch (model/train.py:52-58,113);
* validation = mean loss and mean per-batch PSNR (utils/util.py:200-223);
* PSNR = 20 log10(1 / sqrt(mse)) with +inf below 1e-10 (utils/util.py:69-75);
* a checkpoint is saved when the validation loss OR the PSNR improves
  (model/train.py:115-126), as the dict {epoch, model_state_dict,
  optimizer_state_dict, val_loss, psnr, best_val_loss, best_psnr} plus a JSON
  pointer file {last_epoch, best_model, best_val_loss, best_psnr}; at most
  ``max_checkpoints`` files are kept (utils/util.py:19-64);
* resume reads the pointer file, then the checkpoint it names
  (utils/util.py:78-119);
* low-resolution inputs are the 2x2 average pool of the high-resolution
  target (utils/util.py:313-321).

Nothing here is global: paths and hyper-parameters are arguments.  Batches are
``{"lr": tensor, "hr": tensor}`` dicts (a singleton dim 1 is squeezed, as the
reference's loaders produce it) or ``(lr, hr)`` tuples.
"""
from __future__ import annotations

import glob
import json
import math
import os
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# metrics and data helpers
# ----------------------------------------------------------------------------
def psnr(pred: torch.Tensor, target: torch.Tensor) -> float:
    """Peak signal-to-noise ratio for data in [0, 1]; +inf when mse < 1e-10."""
    mse = torch.mean((target.float() - pred.float()) ** 2).item()
    if mse < 1e-10:
        return float("inf")
    return 20.0 * math.log10(1.0 / math.sqrt(mse))


def downsample2(hr: torch.Tensor) -> torch.Tensor:
    """HR -> LR by 2x2 average pooling (keeps a packed-Bayer channel layout)."""
    return F.avg_pool2d(hr, 2, stride=2)


def synthetic_batches(n_batches: int, batch: int, channels: int, hr_size: int, seed: int = 0,
                      device: str = "cpu") -> Iterator[Dict[str, torch.Tensor]]:
    """Smooth random images in [0, 1] with their 2x down-sampled inputs; stands in for the
    dataset, which is not shipped with the reference."""
    g = torch.Generator().manual_seed(seed)
    for _ in range(n_batches):
        coarse = torch.rand(batch, channels, hr_size // 4, hr_size // 4, generator=g)
        hr = F.interpolate(coarse, size=(hr_size, hr_size), mode="bilinear", align_corners=False).clamp_(0, 1)
        yield {"lr": downsample2(hr).to(device), "hr": hr.to(device)}


def _unpack(batch, device) -> Tuple[torch.Tensor, torch.Tensor]:
    if isinstance(batch, dict):
        lr, hr = batch["lr"], batch["hr"]
    else:
        lr, hr = batch
    lr, hr = lr.to(device).float(), hr.to(device).float()
    if lr.dim() == 5 and lr.shape[1] == 1:
        lr = lr.squeeze(1)
    if hr.dim() == 5 and hr.shape[1] == 1:
        hr = hr.squeeze(1)
    return lr.contiguous(), hr.contiguous()


# ----------------------------------------------------------------------------
# checkpoints
# ----------------------------------------------------------------------------
def save_checkpoint(out_dir: str, log_file: str, model: nn.Module, optimizer: torch.optim.Optimizer, epoch: int,
                    val_loss: float, psnr_db: float, best_val_loss: float, best_psnr: float,
                    max_checkpoints: int = 6) -> str:
    """Write best_model_<time>_epoch<e>_loss<l>_PSNR<p>.pth and the pointer file; prune old files."""
    os.makedirs(out_dir, exist_ok=True)
    stamp = time.strftime("%Y%m%d-%H%M%S")
    path = os.path.join(out_dir, f"best_model_{stamp}_epoch{epoch}_loss{val_loss:.4f}_PSNR{psnr_db:.2f}.pth")
    torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                "val_loss": val_loss, "psnr": psnr_db, "best_val_loss": best_val_loss, "best_psnr": best_psnr}, path)
    with open(log_file, "w") as f:
        json.dump({"last_epoch": epoch, "best_model": path, "best_val_loss": best_val_loss, "best_psnr": best_psnr}, f)
    files = sorted(glob.glob(os.path.join(out_dir, "best_model_*.pth")), key=os.path.getctime)
    for old in files[:max(0, len(files) - max_checkpoints)]:
        try:
            os.remove(old)
        except OSError:
            pass
    return path


def load_checkpoint(log_file: str, model: nn.Module, optimizer: Optional[torch.optim.Optimizer] = None,
                    device: str = "cpu", resume: bool = True) -> Tuple[int, float, float]:
    """Restore model (and optimizer) from the checkpoint the pointer file names.
    Returns (last_epoch, best_val_loss, best_psnr); (0, inf, -inf) when there is nothing to resume.
    Tensors and plain numbers only are read (``weights_only=True``)."""
    fresh = (0, float("inf"), float("-inf"))
    if not resume or not os.path.exists(log_file):
        return fresh
    with open(log_file) as f:
        logs = json.load(f)
    path = logs.get("best_model")
    if not path or not os.path.exists(path):
        return fresh
    ckpt = torch.load(path, map_location=device, weights_only=True)
    model.load_state_dict(ckpt["model_state_dict"])
    if optimizer is not None and "optimizer_state_dict" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    return int(logs.get("last_epoch", 0)), float(logs.get("best_val_loss", float("inf"))), float(logs.get("best_psnr", float("-inf")))


# ----------------------------------------------------------------------------
# model factory and the PSNR evaluation script
# ----------------------------------------------------------------------------
def define_model(name: str = "restormer", device: Optional[str] = None, **kwargs) -> nn.Module:
    """Counterpart of the reference's `define_Model()` (utils/util.py:143-155) without the global `Config`:
    "restormer" builds `Restormer(**kwargs)` -- the reference's defaults (4 channels, dim 12) when no argument is
    given -- on `device` (the GPU when there is one, as the reference picks).  The reference's other choice,
    "unet", is a different network that this package does not contain; as there, an unknown name gives None."""
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    if name == "restormer":
        from .model.Restromer import Restormer
        return Restormer(**kwargs).to(device)
    return None


@torch.no_grad()
def test_psnr(model: nn.Module, val_batches: Iterable, device: Optional[str] = None, log_file: Optional[str] = None,
              verbose: bool = True) -> Dict[str, object]:
    """Counterpart of the reference's tests/testpsnr.py:1-74: restore the best checkpoint named by `log_file`
    (when given), switch to eval, run every validation batch through the model and report per SAMPLE the PSNR, MSE
    and L1 against the target, plus the mean PSNR.  The script's own PSNR has no floor: an exact match gives +inf
    (tests/testpsnr.py:36-40).  Returns {"avg_psnr", "samples": [{"psnr", "mse", "l1"}, ...]}."""
    if device is None:
        device = str(next(model.parameters()).device)
    if log_file is not None:
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)      # as the script: only to restore its state
        load_checkpoint(log_file, model, opt, device)
    model.eval()
    samples = []
    for batch in val_batches:
        lr, hr = _unpack(batch, device)
        out = model(lr)
        for i in range(out.shape[0]):
            mse = torch.mean((out[i].float() - hr[i]) ** 2).item()
            l1 = torch.mean((out[i].float() - hr[i]).abs()).item()
            p = float("inf") if mse == 0 else 20.0 * math.log10(1.0 / math.sqrt(mse))
            samples.append({"psnr": p, "mse": mse, "l1": l1})
            if verbose:
                print(f"Sample {len(samples)}: PSNR={p:.2f} dB, MSE={mse:.6f}, L1={l1:.6f}")
    avg = sum(s["psnr"] for s in samples) / len(samples) if samples else float("nan")
    if verbose:
        print(f"Average PSNR on validation set: {avg:.2f} dB")
    return {"avg_psnr": avg, "samples": samples}


test_psnr.__test__ = False          # a script counterpart, not a pytest case


# ----------------------------------------------------------------------------
# loops
# ----------------------------------------------------------------------------
@torch.no_grad()
def validate(model: nn.Module, batches: Iterable, criterion: Callable = F.l1_loss, device: str = "cpu") -> Tuple[float, float]:
    """Mean loss and mean per-batch PSNR over `batches`."""
    was_training = model.training
    model.eval()
    losses, psnrs = [], []
    for batch in batches:
        lr, hr = _unpack(batch, device)
        out = model(lr)
        losses.append(criterion(out, hr).item())
        psnrs.append(psnr(out, hr))
    model.train(was_training)
    if not losses:
        return float("nan"), float("nan")
    return sum(losses) / len(losses), sum(psnrs) / len(psnrs)


def train_one_epoch(model: nn.Module, batches: Iterable, optimizer: torch.optim.Optimizer, criterion: Callable = F.l1_loss,
                    device: str = "cpu", max_grad_norm: float = 1.0) -> float:
    """One pass over `batches`; returns the mean training loss."""
    model.train()
    losses = []
    for batch in batches:
        lr, hr = _unpack(batch, device)
        optimizer.zero_grad()
        loss = criterion(model(lr), hr)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_grad_norm)
        optimizer.step()
        losses.append(loss.item())
    return sum(losses) / max(len(losses), 1)


@dataclass
class FitResult:
    epochs_run: int = 0
    best_val_loss: float = float("inf")
    best_psnr: float = float("-inf")
    history: List[Dict[str, float]] = field(default_factory=list)
    checkpoints: List[str] = field(default_factory=list)


def fit(model: nn.Module, make_train_batches: Callable[[int], Iterable], make_val_batches: Callable[[], Iterable],
        epochs: int, out_dir: str, log_file: Optional[str] = None, lr: float = 1e-4, weight_decay: float = 1e-6,
        loss: str = "l1", device: str = "cpu", resume: bool = True, max_checkpoints: int = 6) -> FitResult:
    """The reference's training run: Adam + cosine schedule, validation every epoch, a checkpoint whenever
    the validation loss or the PSNR improves.  `make_train_batches(epoch)` / `make_val_batches()` return
    fresh iterables each time they are called."""
    log_file = log_file or os.path.join(out_dir, "logs.json")
    criterion = {"l1": F.l1_loss, "mse": F.mse_loss}[loss]
    model = model.to(device).float()
    optimizer = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=epochs, eta_min=1e-6)
    start, best_loss, best_psnr = load_checkpoint(log_file, model, optimizer, device, resume)
    res = FitResult(best_val_loss=best_loss, best_psnr=best_psnr)
    for epoch in range(start, epochs):
        tr = train_one_epoch(model, make_train_batches(epoch), optimizer, criterion, device)
        vl, vp = validate(model, make_val_batches(), criterion, device)
        scheduler.step()
        res.history.append({"epoch": epoch + 1, "train_loss": tr, "val_loss": vl, "psnr": vp,
                            "lr": optimizer.param_groups[0]["lr"]})
        res.epochs_run += 1
        if vl < res.best_val_loss or vp > res.best_psnr:
            res.best_val_loss = min(res.best_val_loss, vl)
            res.best_psnr = max(res.best_psnr, vp)
            res.checkpoints.append(save_checkpoint(out_dir, log_file, model, optimizer, epoch + 1, vl, vp,
                                                   res.best_val_loss, res.best_psnr, max_checkpoints))
    return res
