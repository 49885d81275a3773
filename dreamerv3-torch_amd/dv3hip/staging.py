"""Host batch -> HBM staging for the training update.

The reference expands the uint8 replay images to float32 on the host and copies 50 MB per cfg-2 batch
synchronously from pageable memory (models.py:176-180).  Here the batch crosses PCIe as it is stored --
uint8 images (12.6 MB), bool flags as one byte -- from PINNED buffers, double-buffered (the host fills batch i+1
while the update on batch i runs); /255, -0.5 and the float conversion happen inside the first kernels that read
the data (dv3_image_to_f32 / dv3_mse_image).

The uploads run on a copy stream of their own beside the previous update (overlap=True, r03).  r02 measured exactly that
SLOWER than uploading in front of the update on its own stream (19.49 against 18.45 ms per update; 18.13 with the batch
resident) and kept the uploads on the update's stream; r03 found the cause -- not the second queue as such but a BLOCKED
one: the copy stream waited on the event that frees its buffer pair, and a queue whose head is a blocked barrier packet
costs every dependent launch of the other queue ~1.3 us ("Compute-unit lanes").  With the host
waiting for that event instead (it has normally fired long before), the copy queue is never blocked and the upload is
free: 16.33 ms per update with a fresh host batch every step against 16.31 with the batch resident (16.57-16.62 with the
upload in front of the update).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch


class BatchStager:
    """stage(host_batch) -> dict of device tensors (valid on the CURRENT stream once returned).

    depth pinned/device buffer pairs are cycled; a buffer pair is reused only after the consumer's stream has
    passed the point where the previous occupant was handed out (recorded with an event), so a captured graph
    may still be reading batch i while batch i+1 is in flight."""

    def __init__(self, device, depth: int = 2, overlap: bool = True):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchStager stages into HBM: it needs a GPU device")
        self.depth = depth
        self._slots: List[Dict[str, tuple]] = [dict() for _ in range(depth)]
        self._free = [None] * depth  # event on the consumer stream: the slot's previous batch has been consumed
        self._h2d = [None] * depth   # event on the copy stream: the slot's previous upload has finished
        self._last = None
        self._i = 0
        # overlap=False: uploads go onto the consumer's own stream (serialised in front of the update)
        self._copy = torch.cuda.Stream(self.device) if overlap else None

    @staticmethod
    def _wire_dtype(k: str, v: np.ndarray):
        if k == "image":
            return torch.uint8
        if v.dtype == np.bool_:
            return torch.uint8
        return torch.float32

    def stage(self, batch: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
        cur = torch.cuda.current_stream(self.device)
        if self._last is not None:
            # whatever consumes the previously returned batch has been queued on `cur` by now: its slot may be
            # overwritten once the stream passes this point
            rel = torch.cuda.Event()
            rel.record(cur)
            self._free[self._last] = rel
        slot = self._slots[self._i]
        copy = self._copy if self._copy is not None else cur
        if self._free[self._i] is not None and copy is not cur:
            # the slot's previous batch (two stage() calls back) must have been consumed.  The HOST waits for that, not
            # the copy stream: a copy queue blocked on the event sits beside the running update's dependent launches and
            # costs each of them ~1.3 us (r02: 19.49 ms per update against 18.45) -- an unblocked one costs nothing.
            self._free[self._i].synchronize()
        if self._h2d[self._i] is not None:
            self._h2d[self._i].synchronize()  # the pinned side is rewritten below: its last upload must be over
        out = {}
        with torch.cuda.stream(copy):
            for k, v in batch.items():
                v = np.asarray(v)
                wd = self._wire_dtype(k, v)
                ent = slot.get(k)
                if ent is None or tuple(ent[0].shape) != v.shape or ent[0].dtype != wd:
                    ent = (torch.empty(v.shape, dtype=wd).pin_memory(), torch.empty(v.shape, dtype=wd, device=self.device))
                    slot[k] = ent
                host, dev = ent
                if k == "image" and v.dtype != np.uint8:
                    v = np.clip(np.rint(v), 0, 255).astype(np.uint8)
                np.copyto(host.numpy(), v, casting="unsafe")
                dev.copy_(host, non_blocking=True)
                out[k] = dev
            done = torch.cuda.Event()
            done.record(copy)
        self._h2d[self._i] = done
        if copy is not cur:
            cur.wait_event(done)
        # flags / scalars become float32 on the device (what the kernels take); images stay uint8
        res = {k: (t if k == "image" else t.to(torch.float32)) for k, t in out.items()}
        self._last = self._i
        self._i = (self._i + 1) % self.depth
        return res
