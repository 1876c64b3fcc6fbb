"""Host-to-HBM batch prefetch on a side HIP stream -- ``data_prefetcher`` of main_ft_mp.py:313-352.

While step i computes on the current stream, batch i+1 is read from the (pinned-memory) loader and copied with
``non_blocking=True`` on a second stream; ``next()`` makes the compute stream wait for that copy, hands the tensors
over (``record_stream`` so the caching allocator does not recycle them under the consumer) and starts the following
copy.  The copy engine (SDMA) runs beside the compute queues, so on MI355X the 38.5 MB/clip-batch transfer hides
entirely behind the step."""
from __future__ import annotations

import torch


class data_prefetcher:
    def __init__(self, loader, opts):
        self.loader = iter(loader)
        self.opts = opts
        self.device = torch.device("cuda", opts.device) if isinstance(opts.device, int) else torch.device(opts.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.preload()

    def preload(self):
        try:
            self.next_input, self.next_target = next(self.loader)
        except StopIteration:
            self.next_input = None
            self.next_target = None
            return
        with torch.cuda.stream(self.stream):
            self.next_input = self.next_input.to(self.device, non_blocking=True)
            self.next_target = self.next_target.to(self.device, non_blocking=True)

    def next(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        inputs, targets = self.next_input, self.next_target
        if inputs is not None:
            inputs.record_stream(torch.cuda.current_stream(self.device))
        if targets is not None:
            targets.record_stream(torch.cuda.current_stream(self.device))
        self.preload()
        return inputs, targets
