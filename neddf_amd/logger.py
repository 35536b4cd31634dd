"""Scalar log of a training run.

Out of the hot path (SURVEY.md section 2 marks the reference's logger package out of
scope); the trainer only needs somewhere to put one row of scalars per step.  Own
design: `ScalarLog.step(...)` is a context manager that times the step and, on exit,
appends one row to the sink -- TensorBoard when the package is importable, else
`log/scalars.jsonl`.  Scalar names follow the reference's run logs (`loss`, `PSNR`,
`objective/<term>`, durations) so dashboards built for them keep working.
"""
import json
import os
import time
from contextlib import contextmanager
from typing import Dict, Iterator, Optional


class _JsonlSink:
    def __init__(self, log_dir: str) -> None:
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.jsonl")
        self.fh = open(self.path, "a")

    def add(self, iteration: int, row: Dict[str, float]) -> None:
        self.fh.write(json.dumps(dict(row, iteration=iteration)) + "\n")
        self.fh.flush()


class _TensorBoardSink:
    def __init__(self, log_dir: str) -> None:
        from torch.utils.tensorboard import SummaryWriter      # optional dependency
        self.writer = SummaryWriter(log_dir=log_dir)

    def add(self, iteration: int, row: Dict[str, float]) -> None:
        for name, value in row.items():
            self.writer.add_scalar(name, value, iteration)


class _NullSink:
    def add(self, iteration: int, row: Dict[str, float]) -> None:
        pass


class StepRecord:
    """What one training step reports; filled by the trainer inside `with log.step() as rec:`."""

    def __init__(self) -> None:
        self.loss: float = float("nan")
        self.psnr: float = float("nan")
        self.terms: Dict[str, float] = {}

    def report(self, loss: float, psnr: float, terms: Dict[str, object]) -> None:
        self.loss, self.psnr = float(loss), float(psnr)
        self.terms = {k: float(v.item() if hasattr(v, "item") else v) for k, v in terms.items()}


class ScalarLog:
    def __init__(self, log_dir: str = "log", sink: Optional[str] = None) -> None:
        """sink: "tensorboard", "jsonl", "null" (ranks that do not write) or None (= tensorboard if importable, else jsonl)."""
        self.iteration = 0
        self.t_open = time.time()
        self.last: Optional[StepRecord] = None
        if sink == "null":
            self.sink = _NullSink()
        elif sink == "jsonl":
            self.sink = _JsonlSink(log_dir)
        elif sink == "tensorboard":
            self.sink = _TensorBoardSink(log_dir)
        else:
            try:
                self.sink = _TensorBoardSink(log_dir)
            except Exception:
                self.sink = _JsonlSink(log_dir)

    @contextmanager
    def step(self) -> Iterator[StepRecord]:
        rec, t0 = StepRecord(), time.time()
        yield rec
        t1 = time.time()
        row = {"loss": rec.loss, "PSNR": rec.psnr, "iteration duration": t1 - t0, "total duration": t1 - self.t_open}
        row.update({"objective/" + k: v for k, v in rec.terms.items()})
        self.sink.add(self.iteration, row)
        self.last = rec
        self.iteration += 1
