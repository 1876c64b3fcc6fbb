"""Batches of a host DataLoader, handed over already resident in HBM.

The fine-tune and validation loops of the reference overlap the host-to-device copy of batch i+1 with the compute of
batch i (main_ft_mp.py:313-352 does it with a hand-driven preload/next object).  Here that is a plain Python generator:

    for inputs, targets in DeviceBatches(loader, device):
        ...

One copy stream per generator; every batch carries the HIP event recorded behind its copies, and the consumer stream
waits on THAT event (not on the whole copy stream), so the copy of the batch after next may already be queued while the
consumer picks this one up.  The copy engine (SDMA) runs beside the compute queues: on MI355X the 38.5 MB of a
16-clip batch hide entirely behind a step."""
from __future__ import annotations

from typing import Iterable, Iterator, Tuple

import torch


def _to_device(obj, device, stream):
    if torch.is_tensor(obj):
        return obj.to(device, non_blocking=True)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_device(o, device, stream) for o in obj)
    return obj


def _record(obj, stream):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)       # the caching allocator must not recycle it under the consumer
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record(o, stream)


def DeviceBatches(loader: Iterable, device, depth: int = 1) -> Iterator[Tuple]:
    """Yield the loader's batches as device tensors, keeping ``depth`` later batches in flight on a copy stream."""
    device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    copy_stream = torch.cuda.Stream(device=device)
    it = iter(loader)
    pending = []                             # [(batch on device, event behind its copies)]

    def enqueue() -> bool:
        try:
            host = next(it)
        except StopIteration:
            return False
        with torch.cuda.stream(copy_stream):
            dev = _to_device(host, device, copy_stream)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        pending.append((dev, ev))
        return True

    for _ in range(max(int(depth), 1)):
        if not enqueue():
            break
    while pending:
        batch, ev = pending.pop(0)
        consumer = torch.cuda.current_stream(device)
        consumer.wait_event(ev)
        _record(batch, consumer)
        enqueue()                            # the next copy runs while the caller computes on this batch
        yield batch
