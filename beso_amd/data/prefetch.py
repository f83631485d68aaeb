"""Host -> device feed of the training step (second half).

The reference's loaders yield dict batches of CPU tensors (``TrajectorySlicerDataset``:
``trajectory_loader.py:160-197``; keys such as ``observation``, ``action``, ``goal_observation``) and the
agent moves them with a blocking ``.to(device)`` inside the step (``obs_encoder.py:17-20``).  At config 3's
8192 samples per step (1024 per GPU) that is 1.4 MB per step per GPU: small, but a *synchronous* copy from
pageable memory stalls the stream once per step.  ``DevicePrefetcher`` wraps any iterable of such batches:
batch k+1 is staged into pinned host buffers (reused, one set per slot) and copied on a side stream while
step k runs; the consumer stream waits on an event, never on the host.  On a CPU device it is a passthrough.
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator

import torch


class DevicePrefetcher:
    def __init__(self, loader: Iterable[dict], device, depth: int = 2):
        self.loader = loader
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        self._cuda = self.device.type == "cuda"
        self._stream = torch.cuda.Stream(self.device) if self._cuda else None
        self._pinned = [dict() for _ in range(self.depth + 1)]      # slot -> key -> pinned staging tensor
        self._copied = [None] * (self.depth + 1)                    # slot -> event: its pinned -> device copies are done

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch: dict, slot: int):
        """Issue the copies of one batch on the side stream; returns (device batch, ready event)."""
        if not self._cuda:
            return batch, None
        out: Dict[str, object] = {}
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()                         # the slot's previous copies have left the pinned buffers
        with torch.cuda.stream(self._stream):
            for key, value in batch.items():
                if not torch.is_tensor(value):
                    out[key] = value
                    continue
                if value.is_cuda:
                    out[key] = value
                    continue
                stage = self._pinned[slot].get(key)
                if stage is None or stage.shape != value.shape or stage.dtype != value.dtype:
                    stage = torch.empty(value.shape, dtype=value.dtype, pin_memory=True)
                    self._pinned[slot][key] = stage
                stage.copy_(value)                                   # pageable -> pinned (host memcpy)
                out[key] = stage.to(self.device, non_blocking=True)  # pinned -> device, asynchronous
            ready = torch.cuda.Event()
            ready.record(self._stream)
        self._copied[slot] = ready
        return out, ready

    def __iter__(self) -> Iterator[dict]:
        it = iter(self.loader)
        queue = []
        slot = 0
        try:
            while len(queue) < self.depth:
                queue.append(self._stage(next(it), slot))
                slot = (slot + 1) % (self.depth + 1)
        except StopIteration:
            pass
        while queue:
            batch, ready = queue.pop(0)
            if ready is not None:
                torch.cuda.current_stream(self.device).wait_event(ready)
                for v in batch.values():                             # the consumer stream now owns the tensors
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(torch.cuda.current_stream(self.device))
            try:
                queue.append(self._stage(next(it), slot))
                slot = (slot + 1) % (self.depth + 1)
            except StopIteration:
                pass
            yield batch
