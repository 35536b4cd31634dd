"""Training observability -- host-side mirror of neddf/logger/{base_logger,nerf_tb_logger}.py.

Same call protocol (write_batchstart / write / write_batchend / next) and the same
scalar names.  TensorBoard is used when the package is importable; otherwise the
scalars go to `log/scalars.jsonl`, one JSON object per iteration."""
import json
import os
from abc import ABC, abstractmethod
from time import time
from typing import Dict

from torch import Tensor


class BaseLogger(ABC):
    """base_logger.py:8-78"""

    def __init__(self) -> None:
        self.reset()

    def reset(self) -> None:
        self.loss: float = 0.0
        self.psnr: float = 0.0
        self.loss_dict: Dict[str, float] = {}
        self.niter: int = 0
        self.loggerstart: float = time()
        self.batchstart = self.prev_batchend = self.batchend = self.loggerstart

    def write(self, loss: float, psnr: float, loss_dict: Dict[str, Tensor]) -> None:
        self.loss, self.psnr = loss, psnr
        self.loss_dict = {key: float(loss_dict[key].item()) for key in loss_dict}

    def write_batchstart(self) -> None:
        self.prev_batchend = self.batchend
        self.batchstart = time()

    def write_batchend(self) -> None:
        self.batchend = time()

    def next(self) -> None:
        log_dict: Dict[str, float] = {"loss": self.loss, "PSNR": self.psnr,
                                      "iteration duration": self.batchend - self.batchstart,
                                      "total duration": self.batchend - self.loggerstart}
        for key in self.loss_dict:
            log_dict["objective/{}".format(key)] = self.loss_dict[key]
        self._next_impl(log_dict)
        self.niter += 1

    @abstractmethod
    def _next_impl(self, data: Dict) -> None:
        raise NotImplementedError()


class NeRFTBLogger(BaseLogger):
    """nerf_tb_logger.py:8-28: scalars under ./log (the run directory)."""

    def __init__(self, log_dir: str = "log") -> None:
        super().__init__()
        self.writer = None
        self.file = None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer = SummaryWriter(log_dir=log_dir)
        except Exception:       # tensorboard is an optional dependency of torch
            os.makedirs(log_dir, exist_ok=True)
            self.file = open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def _next_impl(self, data: Dict) -> None:
        if self.writer is not None:
            for k in data:
                self.writer.add_scalar(k, data[k], self.niter)
        else:
            self.file.write(json.dumps(dict(data, iteration=self.niter)) + "\n")
            self.file.flush()
