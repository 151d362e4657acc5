"""Park the storage of idle tensors in pinned host memory and bring it back.

The reference offloads idle models of the worker runtime to host memory (``OffloadHook`` in
``realhf/api/core/dfg.py``, executed by ``realhf/system/model_worker.py`` after a model function call). Here
a model is ONE flat parameter buffer (plus the fp32 gradient buffer and the optimizer shards), so offloading is
a handful of large copies: every distinct storage is copied to a pinned host buffer (allocated once, reused) and
then shrunk to zero bytes; parameters, `.main_grad` views and optimizer tensors keep their shapes and offsets
and become valid again when the storage is grown back and refilled. With 288 GB per MI355X this is off by
default; it is for layouts that put several large models on one GPU.
"""
from typing import Dict, Iterable, List, Optional, Tuple

import torch


class ParkedStorages:
    def __init__(self):
        self._host: Dict[int, torch.Tensor] = {}  # key: id of the anchor tensor
        self._anchors: List[Tuple[torch.Tensor, int]] = []  # (tensor covering the storage, bytes)
        self.parked = False

    def park(self, tensors: Iterable[Optional[torch.Tensor]], keep_values: Iterable[bool] = ()) -> int:
        """Free the storages behind `tensors` (views of one storage count once); returns the bytes freed.
        `keep_values[i]` False: the content is scratch (e.g. a gradient buffer that is zeroed before its next
        use) and is not copied out; it comes back zero-filled."""
        assert not self.parked, "already parked"
        keep = list(keep_values)
        seen, freed = set(), 0
        for i, t in enumerate(tensors):
            if t is None:
                continue
            st = t.untyped_storage()
            if st.data_ptr() in seen or st.size() == 0:
                continue
            seen.add(st.data_ptr())
            nbytes = st.size()
            host = None
            if not keep or i >= len(keep) or keep[i]:
                host = self._host.get(id(t))
                if host is None or host.numel() != nbytes:
                    host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=t.is_cuda)
                    self._host[id(t)] = host
                flat = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st, 0, (nbytes,))
                host.copy_(flat, non_blocking=False)
            st.resize_(0)
            self._anchors.append((t, nbytes, host))
            freed += nbytes
        self.parked = True
        return freed

    def unpark(self) -> int:
        if not self.parked:
            return 0
        back = 0
        for t, nbytes, host in self._anchors:
            st = t.untyped_storage()
            st.resize_(nbytes)
            flat = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st, 0, (nbytes,))
            if host is not None:
                flat.copy_(host, non_blocking=False)
            else:
                flat.zero_()
            back += nbytes
        self._anchors.clear()
        self.parked = False
        return back
