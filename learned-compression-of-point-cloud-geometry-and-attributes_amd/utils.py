"""Small host-side helpers mirroring the reference's utils.py functions that sit on the path."""


def count_bits(strings):
    """Total bits of a nested list of byte strings (reference: utils.py:30-51; bpp = count_bits / N,
    train.py:268)."""
    total = 0
    for s in strings:
        total += count_bits(s) if isinstance(s, list) else len(s) * 8
    return total


def sparse_collate(coords, feats, device=None):
    """ME.utils.collation.sparse_collate (train.py:185-187): lists of per-sample coordinates [N_i, 3] and
    features [N_i, C] -> (int32 [sum N_i, 4] with the sample index in column 0, float32 [sum N_i, C])."""
    import torch
    cs, fs = [], []
    for b, (c, f) in enumerate(zip(coords, feats)):
        c = torch.as_tensor(c)
        f = torch.as_tensor(f)
        if c.shape[0] != f.shape[0]:
            raise ValueError("sample %d: %d coordinates for %d feature rows" % (b, c.shape[0], f.shape[0]))
        cs.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=torch.int32), torch.floor(c.double()).to(torch.int32)], dim=1))
        fs.append(f.to(torch.float32))
    C, F = torch.cat(cs, dim=0), torch.cat(fs, dim=0)
    if device is not None:
        C, F = C.to(device), F.to(device)
    return C, F


class Prefetcher:
    """Batches built ahead of the training step on a background thread — the role the DataLoader workers play in
    the reference's loop (train.py:171-187): ``make()`` returns one host batch (any object; tensors are pinned when
    ``pin`` is set so that the consumer's ``.to(device, non_blocking=True)`` is an asynchronous copy), ``depth``
    batches are kept ready.  ``make`` is only ever called from the one worker thread, in order, so a seeded
    generator inside it yields the same sequence as a plain loop.  Exceptions raised by ``make`` surface in
    ``next()``."""

    def __init__(self, make, depth=2, pin=True):
        import queue
        import threading
        self._make, self._pin = make, pin
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, name="pcc-batch-prefetch", daemon=True)
        self._th.start()

    def _pinned(self, obj):
        import torch
        if torch.is_tensor(obj):
            return obj.pin_memory() if self._pin and torch.cuda.is_available() and not obj.is_cuda else obj
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._pinned(o) for o in obj)
        return obj

    def _run(self):
        while not self._stop.is_set():
            try:
                item = (self._pinned(self._make()), None)
            except BaseException as e:          # handed to the consumer
                item = (None, e)
            while not self._stop.is_set():
                try:
                    self._q.put(item, timeout=0.1)
                    break
                except Exception:
                    continue
            if item[1] is not None:
                return

    def __iter__(self):
        return self

    def __next__(self):
        item, err = self._q.get()
        if err is not None:
            raise err
        return item

    def close(self):
        self._stop.set()
        self._th.join(timeout=5)
