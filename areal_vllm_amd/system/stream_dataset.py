"""Dataset fed by rollout workers: a background thread pulls trajectory messages from the push/pull
stream and the trainer side drains whatever has arrived so far. Its length is the size of the prompt
dataset the rollout workers iterate (needed for epoch accounting). Behaviour of
``realhf/system/stream_dataset.py:23-106``.
"""
import queue
import threading
from typing import List, Optional

from ..utils import logging
from .data_api import SequenceSample
from .streams import NameResolvingPuller

logger = logging.getLogger("stream-dataset")


class PullerStreamDataset:
    def __init__(self, experiment_name: str, trial_name: str, puller_index: int, dataset_size: int,
                 pull_timeout_ms: int = 100, max_buffered: int = 65536):
        self.dataset_size = dataset_size
        self.data_queue: "queue.Queue[SequenceSample]" = queue.Queue(maxsize=max_buffered)
        self._stop = threading.Event()
        self.pull_timeout_ms = pull_timeout_ms
        self.puller = NameResolvingPuller(experiment_name, trial_name, puller_index)
        self.thread = threading.Thread(target=self._pull_loop, daemon=True, name="stream-puller")
        self.thread.start()

    def _pull_loop(self):
        while not self._stop.is_set():
            try:
                msg = self.puller.pull(timeout_ms=self.pull_timeout_ms)
            except queue.Empty:
                continue
            except Exception as e:  # a broken message must not kill the puller
                logger.error(f"stream puller failed on a message: {e!r}")
                continue
            for d in msg:
                self.data_queue.put(SequenceSample.from_json_compatible(d))

    def __len__(self):
        return self.dataset_size

    def get(self, max_items: Optional[int] = None) -> List[SequenceSample]:
        """Everything that has arrived (up to `max_items`), without waiting."""
        out: List[SequenceSample] = []
        while max_items is None or len(out) < max_items:
            try:
                out.append(self.data_queue.get_nowait())
            except queue.Empty:
                break
        return out

    def close(self):
        self._stop.set()
        self.thread.join(timeout=2)
        self.puller.close()
