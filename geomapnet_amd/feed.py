"""`DeviceFeed`: the host-to-device leg of the reference's input path, one batch ahead of the step.

The reference loads batches through a DataLoader with `pin_memory=True` (/root/reference/common/train.py:180-188)
and issues `data_var.cuda(async=True)` / `target_var.cuda(async=True)` inside `step_feedfwd` (:341,347): the copy
of batch k is asynchronous to the HOST but sits on the step's own stream, in front of the first kernel of step k.
At 192 images a step that copy is 201 MB (fp32 NCHW) or 50 MB (uint8 NHWC, `PoseNet.set_input_u8`) over PCIe
Gen5 x16 (~55 GB/s from pinned memory): 3.7 / 0.9 ms in front of a 13-19 ms step.

`DeviceFeed(loader, device)` wraps any iterable of CPU batches (tuples / lists of tensors) and yields the same
batches as DEVICE tensors whose copies were issued on a copy stream of its own while the PREVIOUS step was
running, into `depth` rotating staging buffers:

    for data, target in DeviceFeed(train_loader, device):
        loss, _ = step_feedfwd(data, model, True, target, criterion, optim, True)   # data.to(dev) is a no-op now

Ordering (what makes the rotation race-free; tests/test_gpu_parity.py::test_device_feed_*):
  * the compute stream waits for batch k's copy event before the batch is handed out;
  * the copy into slot s waits for the event recorded on the compute stream when the consumer came back for the
    next batch after the step that read slot s -- i.e. a staging buffer is rewritten only after every kernel of
    the step that consumed it has finished.  With depth = 2 the copy of batch k+1 therefore runs under step k
    (it needs the slot of batch k-1, whose step has finished before step k started executing).
A yielded batch stays valid until `depth - 1` further batches have been drawn.  On a CPU device the wrapper is
a pass-through.  Plumbing only: torch tensors as containers, torch streams / events as the HIP streams / events.
"""
import collections

import torch


class DeviceFeed:
    def __init__(self, loader, device, depth=2):
        if depth < 2:
            raise ValueError("DeviceFeed needs depth >= 2 (one batch in use, one in flight)")
        self.loader = loader
        self.device = torch.device(device)
        self.depth = int(depth)
        self._cuda = self.device.type == "cuda"
        self._stream = torch.cuda.Stream(device=self.device) if self._cuda else None
        self._bufs = {}
        self._free = [None] * self.depth  # event on the compute stream: the step that read this slot has been enqueued
        self.copy_events = None           # set to a list to collect (start, end) timing events of every copy (bench.py)

    def __len__(self):
        return len(self.loader)

    def _stage(self, t, slot, j):
        if not torch.is_tensor(t):
            return t
        key = (slot, j, tuple(t.shape), t.dtype)
        buf = self._bufs.get(key)
        if buf is None:
            for k in [k for k in self._bufs if k[0] == slot and k[1] == j]:  # a batch of another shape (last partial batch)
                del self._bufs[k]
            buf = self._bufs[key] = torch.empty(t.shape, dtype=t.dtype, device=self.device)
        buf.copy_(t, non_blocking=True)
        return buf

    def _issue(self, k, batch):
        slot = k % self.depth
        seq = isinstance(batch, (tuple, list))
        items = list(batch) if seq else [batch]
        with torch.cuda.stream(self._stream):
            if self._free[slot] is not None:
                self._stream.wait_event(self._free[slot])
            if self.copy_events is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(self._stream)
            dev = [self._stage(t, slot, j) for j, t in enumerate(items)]
            done = torch.cuda.Event(enable_timing=self.copy_events is not None)
            done.record(self._stream)
            if self.copy_events is not None:
                self.copy_events.append((e0, done))
        # `items` (the pinned host tensors) must outlive the asynchronous copy: kept until the batch is handed out
        return (tuple(dev) if seq else dev[0]), done, slot, items

    def __iter__(self):
        if not self._cuda:
            for batch in self.loader:
                yield batch
            return
        it = iter(self.loader)
        pending = collections.deque()
        k = 0
        try:
            pending.append(self._issue(k, next(it)))
            k += 1
        except StopIteration:
            return
        while pending:
            dev, done, slot, _host = pending.popleft()
            torch.cuda.current_stream(self.device).wait_event(done)
            yield dev
            # the consumer has enqueued the step that reads `dev`: its slot is free once the compute stream gets here
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._free[slot] = ev
            try:
                pending.append(self._issue(k, next(it)))
                k += 1
            except StopIteration:
                pass
