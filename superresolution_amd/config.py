"""Stand-in for the reference's top-level ``config`` module (``config.py:5-20``: a bare ``Config`` class that
``utils/util.py``, ``model/train.py`` and ``model/Restromer.py:12`` import by that name) for a process that swaps this
package in under the reference's import names:

    sys.modules["config"] = superresolution_amd.config
    sys.modules["model.Restromer"] = superresolution_amd.model.Restromer

Same attribute names and defaults; nothing is printed at import and ``device`` is looked up when it is asked for, so
importing this module never initialises the GPU."""
import os


class _ConfigMeta(type):
    @property
    def device(cls):
        import torch
        return torch.device("cuda" if torch.cuda.is_available() else "cpu")

    @property
    def log_file(cls):
        return os.path.join(cls.out_dir, "logs.json")


class Config(metaclass=_ConfigMeta):
    # optimiser and schedule (utils/util.py:121-139 reads lr, lr_decay, epochs, loss)
    lr = 1e-4
    lr_decay = 1e-6
    epochs = 1000
    loss = "l1"
    batch_size = 2
    # files (utils/util.py:19-119 and data/loader.py read these)
    out_dir = "./results/mambair"
    train_dir = "data/trainPatches"
    val_dir = "data/val"
    Submission_input = "data/Submission_input"
    submission_save_dir = "data/Restomer_results"
    RESUME = True
    model = "restormer"
