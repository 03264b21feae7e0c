"""Background prefetch of host-side batches (the role of tf.data's prefetch / AUTOTUNE in the reference's pipeline,
neurst/data/dataset_utils.py:308-326): record parsing, bucketing and padding run in a worker thread a few batches
ahead of the training loop; the loop only uploads.  Exceptions of the producer re-raise in the consumer."""
import queue
import threading

_END = object()


class Prefetcher(object):
    def __init__(self, iterable, depth=4):
        self._q = queue.Queue(maxsize=max(1, depth))
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, args=(iter(iterable),), daemon=True)
        self._thread.start()

    def _put(self, item):
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self, it):
        try:
            for item in it:
                if not self._put(item):
                    return
            self._put(_END)
        except BaseException as e:  # noqa: BLE001 -- handed to the consumer
            self._put(e)

    def __iter__(self):
        return self

    def __next__(self):
        item = self._q.get()
        if item is _END:
            self._q.put(_END)
            raise StopIteration
        if isinstance(item, BaseException):
            self._q.put(_END)
            raise item
        return item

    def close(self):
        self._stop.set()
