"""Message transports of the worker runtime, built on TCP sockets and name_resolve.

* ``PushPullStream``: many pushers -> one puller per group, JSON frames (trajectories from rollout workers
  to the trainer side). ``NameResolvingPusher`` / ``NameResolvingPuller`` find each other through
  name_resolve; pushers are spread over the pullers in contiguous groups
  (``realhf/system/push_pull_stream.py:19-180``).
* ``NameResolvingRequestClient`` / ``NameResolvingReplyServer``: master <-> model-worker RPC with
  request ids, optional syn/ack handshake before a request is processed and out-of-order replies
  (``realhf/system/request_reply_stream.py:47-423``).

Frames are ``[4-byte big-endian length][payload]``; control payloads are pickled (same trust domain as
the job itself: all peers are processes of one experiment), trajectory payloads are JSON.
"""
import dataclasses
import json
import pickle
import queue
import select
import socket
import struct
import threading
import time
import uuid
from typing import Any, Dict, Hashable, List, Optional, Tuple

from ..utils import logging, name_resolve, names, network

logger = logging.getLogger("streams")
PUBSUB_BARRIER_NAME = "__pubsub_barrier__"


# ------------------------------------------------------------------------------- framing
def send_frame(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack(">I", len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> Optional[bytes]:
    chunks, got = [], 0
    while got < n:
        b = sock.recv(min(1 << 20, n - got))
        if not b:
            return None
        chunks.append(b)
        got += len(b)
    return b"".join(chunks)


def recv_frame(sock: socket.socket) -> Optional[bytes]:
    head = _recv_exact(sock, 4)
    if head is None:
        return None
    return _recv_exact(sock, struct.unpack(">I", head)[0])


class _Acceptor:
    """Listening socket + reader threads that push every received frame into one queue."""

    def __init__(self, host: Optional[str] = None, port: int = 0):
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.host = host or network.gethostip()
        self.sock.bind(("0.0.0.0", port))
        self.sock.listen(512)
        self.port = self.sock.getsockname()[1]
        self.inbox: "queue.Queue[Tuple[socket.socket, bytes]]" = queue.Queue()
        self._closed = threading.Event()
        self._conns: List[socket.socket] = []
        threading.Thread(target=self._accept_loop, daemon=True).start()

    @property
    def address(self) -> str:
        return f"{self.host}:{self.port}"

    def _accept_loop(self):
        self.sock.settimeout(0.2)
        while not self._closed.is_set():
            try:
                c, _ = self.sock.accept()
            except socket.timeout:
                continue
            except OSError:
                return
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._conns.append(c)
            threading.Thread(target=self._read_loop, args=(c,), daemon=True).start()

    def _read_loop(self, c: socket.socket):
        try:
            while not self._closed.is_set():
                f = recv_frame(c)
                if f is None:
                    return
                self.inbox.put((c, f))
        except OSError:
            return

    def close(self):
        self._closed.set()
        for c in self._conns:
            try:
                c.close()
            except OSError:
                pass
        try:
            self.sock.close()
        except OSError:
            pass


def _connect(addr: str, timeout: float = 60.0) -> socket.socket:
    host, port = addr.rsplit(":", 1)
    deadline = time.time() + timeout
    while True:
        try:
            s = socket.create_connection((host, int(port)), timeout=10)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(None)
            return s
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.2)


# ----------------------------------------------------------------------------- push / pull
class Puller:
    def __init__(self, host: Optional[str] = None, port: int = 0, default_timeout_ms: int = 1000):
        self.acc = _Acceptor(host, port)
        self.default_timeout_ms = default_timeout_ms

    @property
    def address(self) -> str:
        return self.acc.address

    def pull(self, timeout_ms: Optional[int] = None):
        """Next JSON message; raises `queue.Empty` after the timeout."""
        t = (self.default_timeout_ms if timeout_ms is None else timeout_ms) / 1000.0
        _, frame = self.acc.inbox.get(timeout=t)
        return json.loads(frame.decode("utf-8"))

    def close(self):
        self.acc.close()


class Pusher:
    def __init__(self, addr: str):
        self.sock = _connect(addr)
        self._lock = threading.Lock()

    def push(self, data) -> None:
        payload = json.dumps(data).encode("utf-8")
        with self._lock:
            send_frame(self.sock, payload)

    def close(self):
        try:
            self.sock.close()
        except OSError:
            pass


def grouping(num_senders: int, num_receivers: int) -> Dict[int, List[int]]:
    """Contiguous, balanced assignment of senders to receivers."""
    groups: Dict[int, List[int]] = {}
    base, extra = divmod(num_senders, num_receivers)
    start = 0
    for r in range(num_receivers):
        size = base + (1 if r < extra else 0)
        groups[r] = list(range(start, start + size))
        start += size
    return groups


class NameResolvingPuller(Puller):
    def __init__(self, experiment_name: str, trial_name: str, puller_index: int, **kw):
        super().__init__(**kw)
        name_resolve.add(names.stream_pullers(experiment_name, trial_name) + f"/{puller_index}", self.address,
                         replace=True)


class NameResolvingPusher(Pusher):
    def __init__(self, experiment_name: str, trial_name: str, pusher_index: int, pusher_cnt: int, timeout: float = 300):
        root = names.stream_pullers(experiment_name, trial_name)
        deadline = time.time() + timeout
        while True:
            pullers = sorted(name_resolve.find_subtree(root))
            if pullers:
                break
            if time.time() > deadline:
                raise TimeoutError("no stream puller registered")
            time.sleep(0.5)
        idx_of = {int(p.rsplit("/", 1)[1]): p for p in pullers}
        groups = grouping(pusher_cnt, len(idx_of))
        mine = next(r for r, members in groups.items() if pusher_index in members)
        super().__init__(name_resolve.get(idx_of[sorted(idx_of)[mine]]))


# -------------------------------------------------------------------------- request / reply
@dataclasses.dataclass
class Payload:
    handler: Any  # destination worker / model identifier
    handle_name: str
    request_id: str = dataclasses.field(default_factory=lambda: uuid.uuid4().hex)
    syn_reply_id: str = dataclasses.field(default_factory=lambda: uuid.uuid4().hex)
    ack_reply_id: str = dataclasses.field(default_factory=lambda: uuid.uuid4().hex)
    no_syn: bool = True
    send_time: float = 0.0
    data: Any = None
    pre_hooks: List[str] = dataclasses.field(default_factory=list)
    pre_hook_data: List[Any] = dataclasses.field(default_factory=list)
    post_hooks: List[str] = dataclasses.field(default_factory=list)
    post_hook_data: List[Any] = dataclasses.field(default_factory=list)


class NoMessage(Exception):
    pass


class NameResolvingReplyServer:
    """Worker side: receives requests addressed to `handler`, replies on the same connection."""

    def __init__(self, experiment_name: str, trial_name: str, stream_name: str, handler: Hashable):
        self.acc = _Acceptor()
        self.handler = handler
        self._reply_sock: Dict[str, socket.socket] = {}
        self._locks: Dict[int, threading.Lock] = {}
        name_resolve.add(names.request_reply_stream(experiment_name, trial_name, f"{stream_name}/{handler}"),
                         self.acc.address, replace=True)

    def poll(self, block: bool = False, timeout: float = 0.0) -> Payload:
        try:
            c, frame = self.acc.inbox.get(block=block or timeout > 0, timeout=timeout if timeout > 0 else None)
        except queue.Empty:
            raise NoMessage()
        p: Payload = pickle.loads(frame)
        self._reply_sock[p.request_id] = c
        if not p.no_syn:  # syn/ack handshake: tell the master we are ready, wait for its go
            self._send(c, Payload(handler="master", handle_name="syn", request_id=p.syn_reply_id))
            while True:
                c2, f2 = self.acc.inbox.get(timeout=600)
                q: Payload = pickle.loads(f2)
                if q.handle_name == "ack" and q.request_id == p.ack_reply_id:
                    break
                self.acc.inbox.put((c2, f2))  # someone else's message: keep it
        return p

    def _send(self, c: socket.socket, p: Payload):
        lock = self._locks.setdefault(id(c), threading.Lock())
        with lock:
            send_frame(c, pickle.dumps(p))

    def post(self, p: Payload) -> None:
        """Reply: `p.request_id` must be the id of the request being answered."""
        c = self._reply_sock.pop(p.request_id)
        self._send(c, p)

    def close(self):
        self.acc.close()


class NameResolvingRequestClient:
    """Master side: posts requests to many handlers and gathers replies by request id."""

    def __init__(self, experiment_name: str, trial_name: str, stream_name: str, handlers: List[Hashable],
                 timeout: float = 300.0):
        self.socks: Dict[Hashable, socket.socket] = {}
        self._locks: Dict[Hashable, threading.Lock] = {}
        self._replies: Dict[str, Payload] = {}
        self._cv = threading.Condition()
        for h in handlers:
            addr = name_resolve.wait(names.request_reply_stream(experiment_name, trial_name, f"{stream_name}/{h}"),
                                     timeout=timeout)
            s = _connect(addr)
            self.socks[h] = s
            self._locks[h] = threading.Lock()
            threading.Thread(target=self._reader, args=(s,), daemon=True).start()

    def _reader(self, s: socket.socket):
        try:
            while True:
                f = recv_frame(s)
                if f is None:
                    return
                p: Payload = pickle.loads(f)
                with self._cv:
                    self._replies[p.request_id] = p
                    self._cv.notify_all()
        except OSError:
            return

    def post(self, p: Payload) -> str:
        p.send_time = time.monotonic()
        with self._locks[p.handler]:
            send_frame(self.socks[p.handler], pickle.dumps(p))
        return p.request_id

    def request(self, handlers: List[Hashable], handle_type: str, datas: Optional[List[Any]] = None,
                no_syn: bool = True, **hooks) -> List[str]:
        datas = datas if datas is not None else [None] * len(handlers)
        payloads = [Payload(handler=h, handle_name=handle_type, data=d, no_syn=no_syn, **hooks)
                    for h, d in zip(handlers, datas)]
        ids = [self.post(p) for p in payloads]
        if not no_syn:  # wait until every destination is ready, then release them together
            self.gather([p.syn_reply_id for p in payloads])
            for p in payloads:
                self.post(Payload(handler=p.handler, handle_name="ack", request_id=p.ack_reply_id))
        return ids

    def poll(self, request_id: str, block: bool = False, timeout: Optional[float] = None) -> Payload:
        with self._cv:
            if request_id not in self._replies:
                if not block:
                    raise NoMessage()
                ok = self._cv.wait_for(lambda: request_id in self._replies, timeout=timeout)
                if not ok:
                    raise TimeoutError(f"no reply for request {request_id}")
            return self._replies.pop(request_id)

    def gather(self, request_ids: List[str], timeout: Optional[float] = None) -> List[Any]:
        out = []
        for r in request_ids:
            d = self.poll(r, block=True, timeout=timeout).data
            if isinstance(d, dict) and d.get("__error__"):
                raise RuntimeError(f"remote handler failed: {d['__error__']}")
            out.append(d)
        return out

    def call(self, handlers: List[Hashable], handle_type: str, datas: Optional[List[Any]] = None, **kw) -> List[Any]:
        return self.gather(self.request(handlers, handle_type, datas, **kw))

    def close(self):
        for s in self.socks.values():
            try:
                s.close()
            except OSError:
                pass
