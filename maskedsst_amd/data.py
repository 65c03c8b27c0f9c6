"""Input-pipeline contract of the pre-training loop.

The reference's datasets (``src/data_enmap.py:146-310``: rasterio GeoTIFF tiles -> standardised
``{"img": float32 [bands, 64, 64]}``) are out of scope; what the hot path needs from them is the
contract of ``pretrain.py:99-107``: a batch of tiles ``[B, bands, 64, 64]``, ONE random
``image_size x image_size`` window per batch, the crop on the device.

``SyntheticCubeLoader`` honours that contract with a pool of standardised random tiles (per band
N(0, 1), what ``StandardizeEnMAP`` produces, ``src/data_enmap.py:454-457``) and keeps the GPU fed:

* the crop is cut on the host (a [B, bands, 8, 8] window is 64x smaller than the tiles, so only
  51 KB per sample cross PCIe instead of 3.3 MB),
* a worker thread gathers batch i+1 into one of ``prefetch + 1`` pinned staging buffers while the GPU
  works on batch i (numpy releases the GIL during the copy),
* the host->device copy is asynchronous (pinned source, issued on the consumer's current stream), so
  the launching thread never blocks on it; a staging buffer is reused only after the copy that read
  it has completed (event).

There is no CPU fallback for the model; the loader itself also runs without a GPU (``device="cpu"``
returns the pinned-or-pageable batch) so that its logic is testable here.
"""
import queue
import threading

import numpy as np
import torch


class SyntheticCubeLoader:
    def __init__(self, batch_size, bands, image_size=8, tile_size=64, pool_tiles=64, steps=None, seed=5,
                 device="cuda", prefetch=2, zero_pad_bands=0):
        """pool_tiles standardised tiles are drawn once (seeded); every batch samples ``batch_size`` of them with
        replacement and one window position, like a shuffled DataLoader followed by the reference's crop.
        zero_pad_bands: trailing all-zero bands (Houston2018: 48 real + 2, ``src/data_houston2018.py:268-269``)."""
        self.B, self.C, self.S, self.TS = batch_size, bands, image_size, tile_size
        self.steps = steps
        self.device = torch.device(device)
        self.rng = np.random.default_rng(seed)
        g = torch.Generator().manual_seed(seed)
        pool = torch.randn(pool_tiles, bands, tile_size, tile_size, generator=g)
        if zero_pad_bands:
            pool[:, bands - zero_pad_bands:] = 0.0
        self.pool = pool.numpy()
        self.cuda = self.device.type == "cuda"
        n_stage = prefetch + 1
        self.stage = [torch.empty(batch_size, bands, image_size, image_size, pin_memory=self.cuda) for _ in range(n_stage)]
        self.events = [None] * n_stage
        self.free = queue.Queue()
        for i in range(n_stage):
            self.free.put(i)
        self.ready = queue.Queue(maxsize=prefetch)
        self._stop = False
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def draw(self):
        """(tile indices [B], window origin (x, y)) of the next batch -- the only random decisions"""
        idx = self.rng.integers(0, self.pool.shape[0], size=self.B)
        if self.S != self.TS:
            x, y = (int(v) for v in self.rng.integers(0, self.TS - self.S, size=2))
        else:
            x, y = 0, 0
        return idx, (x, y)

    def _work(self):
        n = 0
        while not self._stop and (self.steps is None or n < self.steps):
            slot = self.free.get()
            if slot is None:
                return
            ev = self.events[slot]
            if ev is not None:
                ev.synchronize()              # the async copy that last read this staging buffer
            idx, (x, y) = self.draw()
            np.take(self.pool[:, :, x:x + self.S, y:y + self.S], idx, axis=0, out=self.stage[slot].numpy())
            self.ready.put(slot)
            n += 1
        self.ready.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        slot = self.ready.get()
        if slot is None:
            raise StopIteration
        src = self.stage[slot]
        if self.cuda:
            img = torch.empty(src.shape, dtype=src.dtype, device=self.device)
            img.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.events[slot] = ev
        else:
            img = src.clone()
        self.free.put(slot)
        return img

    def close(self):
        self._stop = True
        self.free.put(None)
        # drain so that a blocked put() in the worker can finish
        try:
            while True:
                self.ready.get_nowait()
        except queue.Empty:
            pass
