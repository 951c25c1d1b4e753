"""GPU image processor: the image half of the reference's dual-encoder processor call (models/datasets.py:172-181,
`processor(text=..., images=image, ...)` -> `pixel_values`), i.e. the ViT feature extractor with its defaults
(resize 224x224 PIL-BILINEAR, rescale 1/255, normalize mean = std = 0.5), computed for a whole batch by two HIP kernel
launches (csrc/image.hip) instead of per item on the host.  Bit-identical to PIL + the HF arithmetic (tests/test_image_*).

    proc = GpuImageProcessor()                       # same attribute names as ViTImageProcessor
    pixel_values = proc(list_of_PIL_or_uint8_HWC)["pixel_values"]      # [n, 3, 224, 224] fp32 on the GPU

There is no CPU fallback: without libmmhip.so / a GPU this raises."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _as_rgb_u8(img):
    """PIL image (any mode -> RGB, as the reference's .convert("RGB")) or uint8 array [h, w, 3] -> contiguous uint8 HWC"""
    if hasattr(img, "convert"):
        img = np.asarray(img.convert("RGB"))
    a = np.ascontiguousarray(img)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"expected a PIL image or a uint8 [h, w, 3] array, got {a.dtype} {a.shape}")
    return a


class GpuImageProcessor:
    model_input_names = ["pixel_values"]

    def __init__(self, size=224, rescale_factor=1 / 255, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5), device=None):
        self.size = {"height": int(size), "width": int(size)}
        self.rescale_factor, self.image_mean, self.image_std = rescale_factor, list(image_mean), list(image_std)
        self.do_resize = self.do_rescale = self.do_normalize = True
        self.device = torch.device(device if device is not None else "cuda:0")
        self._lut = None

    def lut(self):
        """float32 [3, 256]: byte -> ((float32)(byte * rescale_factor, float64)) - mean) / std in float32 -- the arithmetic
        of transformers image_transforms.rescale / normalize, evaluated once on the host"""
        if self._lut is None:
            x = (np.arange(256).astype(np.float64) * self.rescale_factor).astype(np.float32)
            m, s = np.asarray(self.image_mean, dtype=np.float32), np.asarray(self.image_std, dtype=np.float32)
            self._lut = torch.from_numpy(((x[None, :] - m[:, None]) / s[:, None]).astype(np.float32)).to(self.device)
            torch.cuda.synchronize(self.device)          # built once; later used from any stream
        return self._lut

    # ---- host half: pack the decoded images and build the resampling plan (no GPU call)
    def pack(self, images, pin=True, alloc=None):
        """-> (packed uint8 [total + 16], plan int32 [words], n).  Host work only (no kernel launch): usable as / inside a
        DataLoader collate_fn; pass pin=False inside worker processes.  alloc(nbytes) -> uint8 tensor | None: where to pack instead of a
        fresh tensor (a slot of the shared pinned ring, SharedImageRing)."""
        lib = _lib.lib()
        arrs = [_as_rgb_u8(im) for im in images]
        n, S = len(arrs), self.size["height"]
        hs = np.asarray([a.shape[0] for a in arrs], dtype=np.int32)
        ws = np.asarray([a.shape[1] for a in arrs], dtype=np.int32)
        sizes = (hs.astype(np.int64) * ws * 3 + 15) // 16 * 16
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64) if n else np.zeros(0, dtype=np.uint64)
        total = int(sizes.sum())
        # +16: the row-staging kernel reads whole 16-byte chunks; pinned directly when a GPU is present (one host copy)
        packed = alloc(total + 16) if alloc is not None else None
        if packed is None:
            packed = torch.empty(total + 16, dtype=torch.uint8, pin_memory=pin and torch.cuda.is_available())
        pk = packed.numpy()
        for a, o in zip(arrs, offs):
            pk[int(o): int(o) + a.size] = a.reshape(-1)
        words = int(lib.mmhip_image_plan_words(n, hs.ctypes.data_as(C.c_void_p), ws.ctypes.data_as(C.c_void_p), S))
        if words == 0:
            raise _lib.MMHipError("image plan: invalid image sizes")
        plan = torch.empty(words, dtype=torch.int32)
        _lib.check(lib.mmhip_image_plan_build(n, offs.ctypes.data_as(C.c_void_p), hs.ctypes.data_as(C.c_void_p), ws.ctypes.data_as(C.c_void_p),
                                              S, C.c_void_p(plan.data_ptr()), words), "image_plan_build")
        return packed, plan, n

    # ---- device half
    def run(self, packed, plan, n, return_u8=False):
        lib = _lib.lib()
        S = self.size["height"]
        if not torch.cuda.is_available():
            raise _lib.MMHipError("GpuImageProcessor needs a GPU (no CPU fallback)")
        if n == 0:
            out = torch.empty(0, 3, S, S, dtype=torch.float32, device=self.device)
            return (out, torch.empty(0, S, S, 3, dtype=torch.uint8, device=self.device)) if return_u8 else out
        pin = lambda t: t if t.is_pinned() else t.pin_memory()
        packed_h, plan_h = pin(packed), pin(plan)
        packed_d = packed_h.to(self.device, non_blocking=True)
        plan_d = plan_h.to(self.device, non_blocking=True)
        tmp = torch.empty(max(int(lib.mmhip_image_plan_tmp_bytes(C.c_void_p(plan_h.data_ptr()))), 16), dtype=torch.uint8, device=self.device)
        out = torch.empty(n, 3, S, S, dtype=torch.float32, device=self.device)
        u8 = torch.empty(n, S, S, 3, dtype=torch.uint8, device=self.device) if return_u8 else None
        _lib.check(lib.mmhip_image_preprocess(_lib.ptr(packed_d), C.c_void_p(plan_h.data_ptr()), _lib.ptr(plan_d), _lib.ptr(self.lut()),
                                              _lib.ptr(out), _lib.ptr(u8), _lib.ptr(tmp), _lib.stream_ptr()), "image_preprocess")
        # the pinned staging buffers must outlive the asynchronous copies: tie them to the output
        out._mmhip_keep = (packed_h, plan_h, packed_d, plan_d, tmp)
        return (out, u8) if return_u8 else out

    def __call__(self, images, return_tensors="pt", **_):
        if not isinstance(images, (list, tuple)):
            images = [images]
        packed, plan, n = self.pack(images)
        return {"pixel_values": self.run(packed, plan, n)}


class RawImageCollate:
    """collate_fn for datasets that yield decoded images (`image`: PIL / uint8 HWC) instead of `pixel_values`: stacks the
    other fields and does the host half of the GPU processor (packing + plan) inside the DataLoader worker."""

    def __init__(self, processor):
        self.proc = processor

    def __call__(self, items):
        from torch.utils.data import default_collate, get_worker_info
        # inside a DataLoader worker process: no pinned allocation there (it would initialise the GPU runtime in a forked child);
        # the prefetcher pins on the main process instead
        packed, plan, n = self.proc.pack([it["image"] for it in items], pin=get_worker_info() is None)
        batch = default_collate([{k: v for k, v in it.items() if k != "image"} for it in items])
        batch["image_packed"], batch["image_plan"], batch["image_count"] = packed, plan, n
        return batch


class SharedImageRing:
    """Decoded images travel from the DataLoader workers to the training process WITHOUT passing through the DataLoader's result queue: a
    ring of fixed-size slots in shared memory, page-locked in the training process (hipHostRegister), written by the workers' collate
    (RingCollate) and read by the asynchronous host-to-device copy as it stands.  Before, a 33 MB batch (64 decoded 480 x 360 posts) was a torch
    tensor handed over through a shared-memory file per batch (created, mapped, passed as a descriptor, unmapped), then copied into a pinned
    staging buffer by the training thread; the loader fed 0.66 of the step (profiles/r03_loader_bench.txt) with neither side out of cores.
    Slots are handed out through a queue of free slot numbers (back-pressure: a worker waits for a slot) and returned by the prefetcher once
    the copy out of them has completed.  Create it BEFORE the DataLoader forks its workers; decoded bytes are exactly what PIL produced
    (no draft-mode decode): bit-exactness of the image leg is untouched.
    How many slots: the loader delivers in order but the workers take slots in whatever order they get to it, so the batch the consumer waits
    for can belong to the one worker without a slot while later batches hold them all -- nobody moves until that worker's timeout (it then
    packs a plain tensor; correct, but a stall).  With at least num_workers * prefetch_factor slots (every batch the loader can have
    outstanding) plus those the consumer holds (DevicePrefetcher: depth + 1) that cannot happen; datasets.loaders_from_data_key
    sizes ring and prefetch_factor accordingly."""

    MAX_OWNERS = 8

    def __init__(self, slots, slot_bytes, register=True, timeout=5.0):
        import multiprocessing
        self.timeout = float(timeout)             # how long a worker waits for a free slot before it packs a plain tensor instead
        self.slots, self.slot_bytes = int(slots), (int(slot_bytes) + 4095) // 4096 * 4096
        self.buf = torch.empty(self.slots * self.slot_bytes, dtype=torch.uint8).share_memory_()
        # lease book in shared memory: meta[o] = epoch counter of owner o (a loader: train / val / test), meta[MAX_OWNERS + slot] = 0 while the slot
        # is free, else (owner + 1) << 32 | the owner's epoch when a worker took it.  A batch that was prefetched for an epoch the consumer abandoned
        # is dropped by the DataLoader (its _reset discards outstanding results) and the slot it names would be gone for good: begin_epoch /
        # recover hand such slots back
        self.meta = torch.zeros(self.MAX_OWNERS + self.slots, dtype=torch.int64).share_memory_()
        self.free = multiprocessing.Queue()
        for i in range(self.slots):
            self.free.put(i)
        self.pinned = False
        import os
        self._owner = os.getpid()
        if register and torch.cuda.is_available():
            rc = torch.cuda.cudart().cudaHostRegister(self.buf.data_ptr(), self.buf.numel(), 0)
            self.pinned = int(rc) == 0
            if not self.pinned:
                import logging
                logging.getLogger(__name__).warning("SharedImageRing: hipHostRegister of %.0f MB failed (rc %s): the ring stays pageable and every batch takes a "
                                                    "pinned staging copy -- slower than the DataLoader queue it replaces; MMHIP_IMAGE_RING=0 turns the ring off",
                                                    self.buf.numel() / 2 ** 20, rc)

    def view(self, slot, nbytes):
        o = slot * self.slot_bytes
        return self.buf[o: o + nbytes]

    def take(self, nbytes, timeout=None, owner=0):
        """(slot, uint8 view) or (None, None) when the batch does not fit a slot or no slot came free in time (the caller packs a plain tensor)"""
        if nbytes > self.slot_bytes:
            return None, None
        try:
            slot = self.free.get(timeout=self.timeout if timeout is None else timeout)
        except Exception:
            return None, None
        self.meta[self.MAX_OWNERS + slot] = ((int(owner) % self.MAX_OWNERS + 1) << 32) | int(self.meta[int(owner) % self.MAX_OWNERS])
        return slot, self.view(slot, nbytes)

    def release(self, slot):
        self.meta[self.MAX_OWNERS + int(slot)] = 0
        self.free.put(int(slot))

    def begin_epoch(self, owner=0):
        """consumer, BEFORE it asks the loader for a new iterator: slots taken from now on carry the new epoch number"""
        self.meta[int(owner) % self.MAX_OWNERS] += 1

    def recover(self, owner=0, held=()):
        """consumer, AFTER the loader has handed out the new iterator (its _reset has then dropped every result of earlier epochs): a slot this
        owner's workers took in an earlier epoch that is neither free nor held by the consumer can never arrive any more -- back to the free queue.
        Returns how many.  (One live iterator per loader, as DataLoader's persistent workers require anyway.)"""
        o = int(owner) % self.MAX_OWNERS
        now, n = int(self.meta[o]), 0
        for slot in range(self.slots):
            v = int(self.meta[self.MAX_OWNERS + slot])
            if v and (v >> 32) == o + 1 and (v & 0xFFFFFFFF) < now and slot not in held:
                self.release(slot)
                n += 1
        if n:
            import logging
            logging.getLogger(__name__).warning("SharedImageRing: %d slot(s) of an abandoned epoch handed back to the workers", n)
        return n

    def close(self):
        import os
        if self.pinned and os.getpid() == self._owner:          # (forked workers hold a copy of this object; the registration is the parent's)
            torch.cuda.cudart().cudaHostUnregister(self.buf.data_ptr())
            self.pinned = False

    def __del__(self):
        # a ring that is dropped while still registered leaves the runtime with a page-locked range whose mapping is gone: later calls
        # fail with "invalid argument" far from the cause (seen in the test suite after an assertion skipped close())
        try:
            self.close()
        except Exception:
            pass


class RingCollate:
    """RawImageCollate writing the packed images into a SharedImageRing slot: the batch that crosses the DataLoader queue carries the slot
    number, the byte count and the (small) resampling plan instead of the image bytes"""

    def __init__(self, processor, ring, owner=0):
        self.proc, self.ring, self.owner = processor, ring, int(owner)      # owner: which loader's batches these are (SharedImageRing lease book)
        self._starved = False         # (per worker process) the last attempt timed out: slots were leaked by an abandoned epoch or the consumer
                                      # stalls -- do not wait again, look once and fall back, until a slot is there again

    def __call__(self, items):
        from torch.utils.data import default_collate, get_worker_info
        got = {}

        def alloc(nbytes):
            slot, view = self.ring.take(nbytes, timeout=0.0 if self._starved else None, owner=self.owner)
            self._starved = slot is None and nbytes <= self.ring.slot_bytes
            got["slot"], got["bytes"] = slot, nbytes
            return view
        packed, plan, n = self.proc.pack([it["image"] for it in items], pin=get_worker_info() is None, alloc=alloc)
        batch = default_collate([{k: v for k, v in it.items() if k != "image"} for it in items])
        if got.get("slot") is not None:
            batch["image_slot"], batch["image_bytes"] = int(got["slot"]), int(got["bytes"])
        else:
            batch["image_packed"] = packed
        batch["image_plan"], batch["image_count"] = plan, n
        return batch


class DevicePrefetcher:
    """Keeps `depth` batches in flight to the GPU: pinned staging + non-blocking copies on a side stream, so the training
    stream never waits on a pageable host-to-device copy (a 38 MB pixel batch otherwise serialises with the step).
    Batches with `image_packed` / `image_plan` (RawImageCollate) are turned into `pixel_values` on the GPU here."""

    _streams = {}        # one copy stream per device for every prefetcher: the caching allocator pools blocks per stream, and a
                         # fresh stream per epoch strands the previous one's cached blocks (reserved memory grew 0.1-0.3 GiB per instance)

    def __init__(self, loader, device, processor=None, depth=2, trim_padding=True, ring=None):
        self.loader, self.device, self.proc, self.depth = loader, torch.device(device), processor, max(1, depth)
        self.ring, self._held = ring, []          # ring slots whose host-to-device copy is still in flight: (slot, event)
        self.ring_batches = self.queue_batches = 0   # how the image bytes arrived: through a ring slot / as a packed tensor in the DataLoader queue
        if ring is not None and ring.slots < self.depth + 1:
            # `depth` staged batches plus the one being assembled hold a slot each; with fewer the workers wait for slots that only come back
            # after a batch has been handed over -- a stall until their timeout
            raise ValueError(f"SharedImageRing of {ring.slots} slots is too small for {self.depth} staged batches")
        key = (self.device.type, self.device.index if self.device.index is not None else torch.cuda.current_device())
        if key not in DevicePrefetcher._streams:
            DevicePrefetcher._streams[key] = torch.cuda.Stream(device=self.device)
        self.stream = DevicePrefetcher._streams[key]
        self.trim_padding = trim_padding

    @staticmethod
    def trim(batch):
        """drop the token columns no post of the batch uses (right padding beyond the longest post, kept to a multiple of
        32): masked keys weigh exp(-inf) = 0 and rows past a post's end are never read by the heads, so
        outputs and gradients are unchanged while the text tower shrinks with the batch's longest post"""
        m = batch.get("attention_mask")
        if m is None or m.is_cuda:
            return batch
        T = m.shape[-1]
        used = m.reshape(-1, T).ne(0).any(dim=0).nonzero()
        t_eff = min(T, max(32, (int(used.max()) + 1 + 31) // 32 * 32)) if used.numel() else min(T, 32)
        if t_eff < T:
            batch = dict(batch)
            for k in ("input_ids", "attention_mask", "token_type_ids"):
                if k in batch and torch.is_tensor(batch[k]) and batch[k].shape[-1] == T:
                    batch[k] = batch[k][..., :t_eff].contiguous()
        return batch

    def __len__(self):
        return len(self.loader)

    _IMAGE_KEYS = ("image_packed", "image_plan", "image_count", "image_slot", "image_bytes")

    def _stage(self, batch):
        p = batch
        batch = {k: v for k, v in p.items() if k not in self._IMAGE_KEYS}      # everything but the images: trimmed and copied as small tensors
        out = {}
        if self.trim_padding:
            batch = self.trim(batch)
        slots = []
        with torch.cuda.stream(self.stream):
            if "image_slot" in p:
                if self.proc is None or self.ring is None:
                    raise ValueError("ring batches need a GpuImageProcessor and the SharedImageRing they were written to")
                slot = int(p["image_slot"])
                slots.append(slot)
                self.ring_batches += 1
                out["pixel_values"] = self.proc.run(self.ring.view(slot, int(p["image_bytes"])), p["image_plan"], int(p["image_count"]))
            elif "image_packed" in p:
                if self.proc is None:
                    raise ValueError("raw-image batches need a GpuImageProcessor")
                self.queue_batches += 1
                out["pixel_values"] = self.proc.run(p["image_packed"], p["image_plan"], int(p["image_count"]))
            for k, v in batch.items():
                if torch.is_tensor(v) and k != "data_id":
                    h = v if v.is_pinned() else v.pin_memory()
                    out[k] = h.to(self.device, non_blocking=True)
                    out[k]._mmhip_keep = h
                else:
                    out[k] = v
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev, slots

    def _reclaim(self):
        """the ring slots of batches already handed to the consumer go back to the workers.  Called before every blocking wait for the next
        batch: a worker may be waiting for exactly these slots (the loader returns batches in order, so the batch awaited can be the one whose
        worker has no slot yet) -- leaving the release to "whenever the next batch has arrived" deadlocks until the worker's timeout.  The
        copies out of these slots were enqueued a batch ago: the host-side wait is over by the time it is asked for."""
        for slot, ev in self._held:
            ev.synchronize()
            self.ring.release(slot)
        self._held = []

    def _ring_owner(self):
        c = getattr(self.loader, "collate_fn", None)
        c = getattr(c, "inner", c)                 # BatchTokenizeCollate wraps the image collate
        return getattr(c, "owner", 0)

    def prime(self):
        """ask the loader for its next epoch's iterator NOW: its workers start on the epoch's first batches while the caller is still busy with
        something else (the tail of the previous epoch on another loader: MMLate_Model.train, MMHIP_EPOCH_PREFETCH).  The sampler's shuffle is
        drawn here.  The following __iter__ continues with this iterator."""
        if getattr(self, "_primed", None) is None:
            if self.ring is not None:
                self.ring.begin_epoch(self._ring_owner())
            self._primed = iter(self.loader)
            if self.ring is not None:              # slots named by batches of an abandoned epoch (dropped by the loader's reset) come back
                self.ring.recover(self._ring_owner(), {slot for slot, _ in self._held})
        return self

    def __iter__(self):
        self.prime()
        it, self._primed = self._primed, None
        queue = []

        def fetch():
            if self.ring is not None:
                self._reclaim()
            queue.append(self._stage(next(it)))       # StopIteration here ends the epoch
        try:
            try:
                while len(queue) < self.depth:
                    fetch()
            except StopIteration:
                pass
            while queue:
                batch, ev, slots = queue.pop(0)
                torch.cuda.current_stream(self.device).wait_event(ev)
                for v in batch.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(torch.cuda.current_stream(self.device))
                self._held += [(slot, ev) for slot in slots]
                yield batch
                try:
                    fetch()
                except StopIteration:
                    pass
        finally:
            # also when the consumer stops early (break, exception): the slots of staged batches go back once their copies are done
            if self.ring is not None:
                self._held += [(slot, ev) for _, ev, slots in queue for slot in slots]
                self._reclaim()
