"""The training / evaluation harness driving the HIP network on the GPU, on files in the reference's on-disk
format: `fit` for two epochs on synthetic `.npz` samples, the checkpoint it writes restored into a fresh model (the
reference's dict layout, utils/util.py:19-64), `test_psnr` (tests/testpsnr.py) on the restored model, and the
dataset item computed on the GPU against the host loader."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fit_checkpoint_and_psnr_script(tmp_path):
    from superresolution_amd import data, harness
    train_dir, val_dir, out_dir = (str(tmp_path / d) for d in ("train", "val", "out"))
    data.write_synthetic_npz(train_dir, 6, size=32, channels=4, seed=1)
    data.write_synthetic_npz(val_dir, 2, size=48, channels=4, seed=2)
    train, val = data.loaders(train_dir, val_dir, batch_size=2)
    torch.manual_seed(0)
    model = harness.define_model("restormer")                       # reference defaults: 4 channels, dim 12
    assert next(model.parameters()).is_cuda and harness.define_model("something else") is None
    res = harness.fit(model, lambda epoch: train, lambda: val, epochs=2, out_dir=out_dir, device="cuda")
    assert res.epochs_run == 2 and res.checkpoints and all(math.isfinite(h["train_loss"]) for h in res.history)
    assert res.history[1]["train_loss"] < res.history[0]["train_loss"] * 1.05
    # the file carries the reference's keys and loads strictly into a fresh module
    ckpt = torch.load(res.checkpoints[-1], map_location="cpu", weights_only=True)
    assert set(ckpt) == {"epoch", "model_state_dict", "optimizer_state_dict", "val_loss", "psnr", "best_val_loss", "best_psnr"}
    fresh = harness.define_model("restormer")
    rep = harness.test_psnr(fresh, val, log_file=os.path.join(out_dir, "logs.json"), verbose=False)
    for (k, a), b in zip(fresh.state_dict().items(), model.state_dict().values()):
        assert torch.equal(a, b), k
    assert len(rep["samples"]) == 2 and all(s["mse"] > 0 and s["l1"] > 0 for s in rep["samples"])
    # per-sample PSNR of the script == the harness' validation PSNR (validation batches hold one sample)
    vl, vp = harness.validate(fresh, val, device="cuda")
    assert abs(rep["avg_psnr"] - vp) < 1e-4
    # resume: nothing left to do at epochs=2, one more epoch at epochs=3
    again = harness.fit(harness.define_model("restormer"), lambda epoch: train, lambda: val, epochs=3, out_dir=out_dir, device="cuda")
    assert again.epochs_run == 1


@pytest.mark.parametrize("dtype,shape", [("u16", (3, 33, 50, 4)), ("f32", (2, 16, 24, 3)), ("u16", (1, 64, 64, 1))])
def test_dataset_item_on_the_gpu_is_bit_identical(tmp_path, dtype, shape):
    from superresolution_amd import data
    rng = np.random.default_rng(sum(shape))
    B, H, W, C = shape
    max_vals = [1023, 4095, 16383][:B] if B <= 3 else [1023] * B
    raws = []
    for b in range(B):
        r = rng.integers(0, max_vals[b] + 1, size=(H, W, C))
        raws.append(r.astype(np.uint16) if dtype == "u16" else (r + rng.random((H, W, C))).astype(np.float32))
        np.savez(str(tmp_path / f"{b:05d}.npz"), raw=raws[-1], max_val=np.array(max_vals[b]))
    ds = data.RawPatchDataset(str(tmp_path))
    items = [ds[b] for b in range(B)]
    raw = torch.from_numpy(np.stack(raws).astype(np.int16 if dtype == "u16" else np.float32))
    if dtype == "u16":
        raw = raw.view(torch.uint16)
    got = data.prepare_batch_gpu(raw.cuda(), max_vals)
    hr = torch.cat([it["hr"] for it in items]).cuda()
    lr = torch.cat([it["lr"] for it in items]).cuda()
    assert got["hr"].shape == hr.shape and got["lr"].shape == lr.shape
    assert torch.equal(got["hr"], hr)
    assert torch.equal(got["lr"], lr)
