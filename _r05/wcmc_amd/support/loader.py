"""Pinned, double-buffered host -> device stager and patch loader (the loader half).

The reference's loader (``train_kpcn.py:177-188``: ``DataLoader(num_workers=1, pin_memory=False)`` around
``MSDenoiseDataset.__getitem__``, ``datasets.py:1026-1146``) preprocesses every patch in numpy on ONE CPU worker and hands
pageable tensors to ``batch[k].cuda()``.  Here the per-image arithmetic runs on the GPU (``support/datasets.py``:
``DenoisePreprocessor``, ``PatchBatcher``), so what is left for the host is to get an image's raw renderer output across
PCIe without stalling the training stream:

  * ``HostReaderPool``: ``workers`` (default 2) reader threads call the user's ``reader(index)`` (file format and I/O are the
    caller's: SURVEY.md 8 keeps the dataset files out of scope) and copy the arrays into a ring of PINNED staging buffers,
    several images at once, handed out in order;
  * ``ImageStager``: a background thread takes the staged images, enqueues the
    host -> device copies and the two preprocessing kernels on a COPY STREAM, and hands over device-resident
    ``(kpcn, llpm, gt, prob)`` behind an event -- image i + 1 crosses PCIe and is preprocessed while the training stream
    is still drawing patches from image i;
  * ``PatchLoader``: the iterable the epoch loop consumes (``dataloaders['train']`` of ``wcmc_amd.train_kpcn``):
    ``patches_per_image`` patches per image in batches of ``batch_size``, origins importance-sampled with the reference's
    ``np.random.choice`` call (``datasets.py:795-810``), one ``wcmc_assemble_kpcn_patches`` launch per batch -- enqueued on a
    side stream by a producer thread ``prefetch`` batches ahead of the consumer, which only makes its stream wait for the
    batch's event.

``scripts/time_loader.py`` measures it (patches/s and PCIe GB/s) next to the train step's consumption rate.
"""
import ctypes
import queue
import threading

import numpy as np
import torch

from .datasets import DenoisePreprocessor, PatchBatcher


class HostReaderPool:
    """The HOST half of the stager: ``workers`` threads call ``reader(index)`` and copy its arrays into a ring of staging buffers
    (pinned when a GPU is present), concurrently, while results are handed out IN ORDER of ``indices``.  The reference's
    ``DataLoader(num_workers=...)`` (``train_kpcn.py:177-188``) is the model: several images are read and staged at once, the
    training thread never touches a file.  No GPU call in here (``tests/test_cpu_host.py`` runs eight of these side by side).

    Iterating yields ``(slot, prob, nbytes)``: ``slot['raw']`` / ``slot['gt']`` hold the staged arrays; give the slot back with
    ``release(slot)`` once its contents have been consumed (after the host -> device copy's event, for a pinned slot).

    Pinned host memory: ``depth + workers`` slots stay allocated for the life of the pool, each the size of the largest image
    read so far -- 0.9 GB for a 512 x 512 x 8 spp frame of 104 raw channels, i.e. 3.6 GB per rank with the defaults (two workers,
    depth two), eight times that on a node with eight ranks.  ``workers=1, depth=1`` halves it; ``pin=False`` stages in pageable
    memory (the copy stream then waits for the driver's own staging)."""

    def __init__(self, reader, indices, workers=2, depth=2, pin=None):
        assert workers >= 1 and depth >= 1
        self.reader, self.indices = reader, list(indices)
        self.workers, self.depth = int(workers), int(depth)
        self.pin = torch.cuda.is_available() if pin is None else bool(pin)
        self.free_q = queue.Queue()
        for _ in range(self.depth + self.workers):                    # `depth` handed out + one being filled per worker
            self.free_q.put({})
        self.stop = threading.Event()

    def release(self, slot):
        self.free_q.put(slot)

    def _buffer(self, slot, key, arr):
        buf = slot.get(key)
        if buf is None or buf.shape != arr.shape:
            buf = torch.empty(arr.shape, dtype=torch.float32, pin_memory=self.pin)
            slot[key] = buf
        return buf

    def _load(self, i):
        slot = ImageStager._get(self.free_q, self.stop)               # a staging slot whose last consumer is done with it
        if slot is None:
            return None
        if slot.get('event') is not None:
            slot['event'].synchronize()                               # (the previous host -> device copy out of this slot)
        item = self.reader(i)
        raw = np.ascontiguousarray(item['raw'], dtype=np.float32)
        gt = np.ascontiguousarray(item['gt'], dtype=np.float32)
        p_raw, p_gt = self._buffer(slot, 'raw', raw), self._buffer(slot, 'gt', gt)
        # pageable -> staging through ctypes: the foreign call runs WITHOUT the interpreter lock.  ``Tensor.copy_`` held it for the
        # whole 0.9 GB memcpy of a 512x512x8-spp image -- the training thread could not enqueue a step meanwhile: +1.4 ms per
        # step (scripts/time_loader.py)
        ctypes.memmove(p_raw.data_ptr(), raw.ctypes.data, raw.nbytes)
        ctypes.memmove(p_gt.data_ptr(), gt.ctypes.data, gt.nbytes)
        return slot, item.get('prob'), raw.nbytes + gt.nbytes

    def __iter__(self):
        import collections
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="wcmc-reader")
        pending, it = collections.deque(), iter(self.indices)
        try:
            for i in self.indices[:self.workers]:
                pending.append(pool.submit(self._load, next(it)))
            while pending:
                res = pending.popleft().result()                      # in order of `indices`; reader errors surface here
                nxt = next(it, None)
                if nxt is not None and not self.stop.is_set():
                    pending.append(pool.submit(self._load, nxt))
                if res is None:
                    return
                yield res
        finally:
            self.stop.set()
            for f in pending:
                f.cancel()
            pool.shutdown(wait=True)


class ImageStager:
    """Iterate ``(kpcn (H,W,44), llpm (H,W,S,37) | None, gt (H,W,9), prob (H,W) numpy | None)`` device buffers of the images
    ``indices``; ``reader(i)`` returns ``{'raw': (H,W,S,C>=104) float32, 'gt': (H,W,9) float32, 'prob': (H,W) | None}`` numpy
    arrays (any object with the buffer protocol that ``torch.from_numpy`` / ``np.asarray`` accepts, e.g. a memmap)."""

    def __init__(self, reader, indices, device, depth=2, use_llpm=True, max_depth=DenoisePreprocessor.MAX_DEPTH, workers=2):
        """workers: reader / staging threads (``HostReaderPool``): images i + 1 .. i + workers are read from disk and copied
        into pinned memory concurrently while image i crosses PCIe."""
        assert depth >= 2, "double buffering needs two staging slots"
        self.workers = max(1, int(workers))
        self.reader, self.indices, self.device = reader, list(indices), torch.device(device)
        if self.device.index is None:                                 # 'cuda' -> the current device, by index (threads need it)
            self.device = torch.device(self.device.type, torch.cuda.current_device())
        self.depth, self.use_llpm = depth, use_llpm
        self.pre = DenoisePreprocessor(max_depth)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.bytes_moved = 0

    @staticmethod
    def _get(q, stop):
        """``q.get()`` that gives up when the consumer has gone (``stop``): the producer must never block forever on a queue
        nobody serves any more -- a daemon thread parked there would keep the pinned ring and the device tensors alive."""
        while not stop.is_set():
            try:
                return q.get(timeout=0.2)
            except queue.Empty:
                continue
        return None

    @staticmethod
    def _put(q, item, stop):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.2)
                return True
            except queue.Full:
                continue
        return False

    def _produce(self, out_q, hostpool, stop):
        try:
            torch.cuda.set_device(self.device)
            for slot, prob, nbytes in hostpool:                       # staged images, in order; several are in flight
                p_raw, p_gt = slot['raw'], slot['gt']
                with torch.cuda.stream(self.copy_stream):
                    d_raw = p_raw.to(self.device, non_blocking=True)
                    d_gt = p_gt.to(self.device, non_blocking=True)
                    kpcn = self.pre._preprocess_kpcn(d_raw)
                    llpm = self.pre._preprocess_llpm(d_raw) if self.use_llpm else None
                    ev = torch.cuda.Event()
                    ev.record(self.copy_stream)
                slot['event'] = ev
                self.bytes_moved += nbytes
                if not self._put(out_q, (kpcn, llpm, d_gt, prob, ev, slot), stop):
                    return
            self._put(out_q, None, stop)
        except BaseException as exc:                                  # surface reader / CUDA errors in the consumer
            self._put(out_q, exc, stop)

    def __iter__(self):
        out_q, stop = queue.Queue(maxsize=self.depth), threading.Event()
        hostpool = HostReaderPool(self.reader, self.indices, workers=self.workers, depth=self.depth, pin=True)
        worker = threading.Thread(target=self._produce, args=(out_q, hostpool, stop), daemon=True)
        worker.start()
        try:
            while True:
                got = out_q.get()
                if got is None:
                    return
                if isinstance(got, BaseException):
                    raise got
                kpcn, llpm, gt, prob, ev, slot = got
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)                                    # the consumer's stream, not the host, waits
                for t in (kpcn, llpm, gt):
                    if t is not None:
                        t.record_stream(cur)
                hostpool.release(slot)                                # (its event guards the pinned buffers' reuse)
                yield kpcn, llpm, gt, prob
        finally:
            stop.set()                                                # the producer's queue waits poll this flag ...
            hostpool.stop.set()
            hostpool.release(None)                                    # ... and a sentinel wakes a reader blocked on the ring
            worker.join(timeout=5.0)


class PatchLoader:
    """Batches of the KPCN base model over the staged images; ``len()`` = batches per epoch."""

    def __init__(self, reader, indices, device, batch_size=8, patch_size=PatchBatcher.PATCH_SIZE, use_llpm=True, depth=2,
                 patches_per_image=None, prefetch=2, workers=2):
        self.stager = ImageStager(reader, indices, device, depth=depth, use_llpm=use_llpm, workers=workers)
        self.batcher = PatchBatcher(patch_size, batch_size)
        if patches_per_image is not None:
            self.batcher.patches_per_image = (patches_per_image // batch_size) * batch_size
        self.batch_size = batch_size
        self.assemble_stream = torch.cuda.Stream(device=self.stager.device)
        self.prefetch = max(1, int(prefetch))                         # batches assembled ahead of the consumer
        # Pacing (``kick``): the producer's Python work for batch t + prefetch competes with the training thread for the
        # interpreter -- and it used to start exactly when the consumer popped batch t, i.e. while the training thread was
        # enqueueing step t (copy-in, pairings, hipGraphLaunch: ~1.2 ms of host work with the GPU idle behind it).  A consumer
        # that calls ``kick()`` once its step is enqueued (``GraphedTrainStep.after_enqueue``) moves that work under the
        # step's GPU time; without kicks the producer proceeds after a short timeout.
        self._tick = threading.Semaphore(0)
        self._paced = False                                           # becomes True with the first kick: un-kicked consumers are not throttled
        self.pace_timeout = 0.05

    def kick(self):
        """The consumer has enqueued its step and is about to wait for it: assemble the next batch now."""
        self._paced = True
        self._tick.release()

    def __len__(self):
        return len(self.stager.indices) * (self.batcher.patches_per_image // self.batch_size)

    def _produce(self, out_q, stop):
        """Producer thread: walks the staged images, draws an image's origins and enqueues the assembly of its batches on
        the side stream; hands ``(batch, event)`` to the consumer through a bounded queue."""
        p = self.batcher.patch_size
        dev = self.stager.device
        side = self.assemble_stream
        images = None
        try:
            torch.cuda.set_device(dev)
            images = iter(self.stager)
            while not stop.is_set():
                with torch.cuda.stream(side):                     # (the stager hands its buffers to the CURRENT stream)
                    got = next(images, None)
                if got is None:
                    break
                kpcn, llpm, gt, prob = got
                h, w = kpcn.shape[:2]
                if prob is None:
                    prob = np.zeros((h, w), dtype=np.float64)         # (not a distribution: uniform, as the reference falls back)
                # origins must keep the window inside the image: the reference crops what it gets, which silently shrinks a
                # patch at the border; its probability maps are zero there (datasets.py:795-810)
                valid = np.zeros((h, w), dtype=np.float64)
                valid[:h - p + 1, :w - p + 1] = np.asarray(prob, dtype=np.float64)[:h - p + 1, :w - p + 1]
                s = valid.sum()
                if s > 0:
                    valid /= s
                else:
                    valid[:h - p + 1, :w - p + 1] = 1.0 / ((h - p + 1) * (w - p + 1))
                origins = self.batcher.sample_origins(valid)
                self.batcher.check_origins(origins, h, w)
                with torch.cuda.stream(side):
                    origins_dev = torch.as_tensor(origins, dtype=torch.int32).to(dev)      # one copy per image
                    for k in range(0, len(origins), self.batch_size):
                        if self._paced and not self._tick.acquire(timeout=self.pace_timeout):
                            self._paced = False           # no kick within the timeout: this consumer does not pace -- stop waiting for it
                        if stop.is_set():
                            return
                        batch = self.batcher.batch(kpcn, llpm, gt, origins_dev[k:k + self.batch_size], check=False)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        if not ImageStager._put(out_q, (batch, ev), stop):
                            return
            ImageStager._put(out_q, None, stop)
        except BaseException as exc:                              # surface reader / CUDA errors in the consumer
            ImageStager._put(out_q, exc, stop)
        finally:
            if images is not None:
                images.close()

    def __iter__(self):
        """Batches are assembled AHEAD of the consumer, on a side stream, by a producer thread: while the training thread
        waits for its step (the wait releases the interpreter lock) the next batches' origins are drawn and their assembly
        kernels enqueued, so between two steps the consumer only pops a queue and makes its stream wait for an event
        (the assembly kernel, 150 us, and ~0.3 ms of host work per batch used to sit in the gap between two graph replays:
        ``scripts/time_loader.py``).  ``numpy.random`` is drawn from on the producer thread, in image order."""
        dev = self.stager.device
        out_q, stop = queue.Queue(maxsize=self.prefetch), threading.Event()
        # every pass starts un-paced, with no permits left over from the last one: pacing belongs to the consumer of THIS pass
        # (a loader used with --graph and then eagerly would otherwise wait 50 ms per batch for kicks that never come)
        self._paced = False
        while self._tick.acquire(blocking=False):
            pass
        worker = threading.Thread(target=self._produce, args=(out_q, stop), daemon=True)
        worker.start()
        try:
            while True:
                got = out_q.get()
                if got is None:
                    return
                if isinstance(got, BaseException):
                    raise got
                batch, ev = got
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(ev)
                seen = set()
                for t in batch.values():                          # (the entries are views of one allocation: one call)
                    if isinstance(t, torch.Tensor):
                        key = t.untyped_storage().data_ptr()
                        if key not in seen:
                            seen.add(key)
                            t.record_stream(cur)
                yield batch
        finally:
            stop.set()
            try:                                                  # wake a producer blocked on a full queue
                while True:
                    out_q.get_nowait()
            except queue.Empty:
                pass
            worker.join(timeout=10.0)
