"""Minimal stand-in for pytorch_lightning (not installed in this image; TEST INFRASTRUCTURE ONLY).

Enough of the 1.x surface for `nndet.ptmodule` (nndet/ptmodule/base_module.py, retinaunet/base.py) to import and for
a LightningModule to be constructed and stepped by hand: `LightningModule` is an `nn.Module` with `log`, `log_dict`,
`print`, `trainer`, `current_epoch`, `global_step`; `Callback` and `Trainer` are inert placeholders."""
import torch

__version__ = "1.4.9-shim"


class LightningModule(torch.nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.trainer = None
        self.current_epoch = 0
        self.global_step = 0
        self.logged = {}

    def log(self, name, value, *args, **kwargs):
        self.logged[name] = value

    def log_dict(self, d, *args, **kwargs):
        self.logged.update(d)

    def print(self, *args, **kwargs):
        print(*args, **kwargs)

    def summarize(self, *args, **kwargs):
        return None

    def on_epoch_start(self):
        return None

    def on_train_epoch_start(self):
        return None

    def validation_epoch_end(self, outputs):
        return None

    def training_epoch_end(self, outputs):
        return None

    def save_hyperparameters(self, *args, **kwargs):
        return None


class Callback:
    pass


class Trainer:
    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs


class LightningDataModule:
    def __init__(self, *args, **kwargs):
        pass
