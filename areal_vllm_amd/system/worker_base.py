"""Worker lifecycle, control channel and liveness for the multi-process runtime.

A worker is configured, started, polled in a loop and can be paused / resumed / told to exit through a
control server that the controller reaches over TCP (address published in name_resolve). Its status is
mirrored into name_resolve with a keep-alive TTL so that peers and the controller notice a dead worker;
a worker in turn watches the keys of the workers it depends on and exits when one of them disappears.
Capabilities of ``realhf/system/worker_base.py`` + ``worker_control.py`` (ZMQ there, sockets here).
"""
import asyncio
import dataclasses
import enum
import pickle
import queue
import threading
import time
import traceback
from typing import Any, Callable, Dict, List, Optional

from ..utils import logging, name_resolve, names
from .streams import _Acceptor, _connect, recv_frame, send_frame

logger = logging.getLogger("worker")
WORKER_KEEPALIVE_TTL = 15.0


class WorkerServerStatus(str, enum.Enum):
    READY = "READY"
    RUNNING = "RUNNING"
    PAUSED = "PAUSED"
    COMPLETED = "COMPLETED"
    ERROR = "ERROR"
    UNKNOWN = "UNKNOWN"
    LOST = "LOST"
    INTERRUPTED = "INTERRUPTED"


class ExpStatus(str, enum.Enum):
    RUNNING = "RUNNING"
    ABORTED = "ABORTED"
    COMPLETE = "COMPLETE"


class WorkerException(Exception):
    def __init__(self, worker_name, worker_status, scenario):
        super().__init__(f"Worker {worker_name} is {worker_status} while {scenario}")
        self.worker_name, self.worker_status = worker_name, worker_status


@dataclasses.dataclass
class PollResult:
    sample_count: int = 0
    batch_count: int = 0


@dataclasses.dataclass
class WorkerInformation:
    experiment_name: str = ""
    trial_name: str = ""
    worker_type: str = ""
    worker_index: int = -1
    worker_count: int = 0
    worker_tag: Optional[str] = None
    host_key: Optional[str] = None
    watch_keys: Optional[List[str]] = None

    @property
    def worker_name(self) -> str:
        return f"{self.worker_type}/{self.worker_index}"


class WorkerServer:
    """Control endpoint of one worker: named commands -> handler functions, plus status publishing."""

    def __init__(self, experiment_name: str, trial_name: str, worker_name: str):
        self.experiment_name, self.trial_name, self.worker_name = experiment_name, trial_name, worker_name
        self.acc = _Acceptor()
        self.handlers: Dict[str, Callable] = {}
        name_resolve.add(names.worker(experiment_name, trial_name, *worker_name.split("/")), self.acc.address,
                         replace=True, keepalive_ttl=WORKER_KEEPALIVE_TTL)
        self.set_status(WorkerServerStatus.READY)

    def register_handler(self, command: str, fn: Callable):
        self.handlers[command] = fn

    def set_status(self, status: WorkerServerStatus):
        name_resolve.add(names.worker_status(self.experiment_name, self.trial_name, self.worker_name), status.value,
                         replace=True, delete_on_exit=False)

    def handle_requests(self, max_count: Optional[int] = None) -> int:
        n = 0
        while max_count is None or n < max_count:
            try:
                c, frame = self.acc.inbox.get_nowait()
            except queue.Empty:
                break
            cmd, kwargs = pickle.loads(frame)
            try:
                if cmd not in self.handlers:
                    raise KeyError(f"unknown worker command {cmd}")
                res = (True, self.handlers[cmd](**kwargs))
            except Exception as e:  # surfaced to the controller
                logger.error(f"worker command {cmd} failed: {traceback.format_exc()}")
                res = (False, repr(e))
            send_frame(c, pickle.dumps(res))
            n += 1
        return n

    def close(self):
        self.acc.close()


class WorkerControlPanel:
    """Controller side: connect to named workers, broadcast commands, query status."""

    def __init__(self, experiment_name: str, trial_name: str, timeout: float = 300.0):
        self.experiment_name, self.trial_name, self.timeout = experiment_name, trial_name, timeout
        self.socks: Dict[str, Any] = {}

    def connect(self, worker_names: List[str], timeout: Optional[float] = None) -> List[str]:
        for w in worker_names:
            if w in self.socks:
                continue
            addr = name_resolve.wait(names.worker(self.experiment_name, self.trial_name, *w.split("/")),
                                     timeout=timeout or self.timeout)
            self.socks[w] = _connect(addr)
        return list(self.socks)

    def request(self, worker_name: str, command: str, **kwargs):
        s = self.socks[worker_name]
        send_frame(s, pickle.dumps((command, kwargs)))
        f = recv_frame(s)
        if f is None:
            raise WorkerException(worker_name, WorkerServerStatus.LOST, f"handling {command}")
        ok, res = pickle.loads(f)
        if not ok:
            raise WorkerException(worker_name, WorkerServerStatus.ERROR, f"handling {command}: {res}")
        return res

    def group_request(self, command: str, worker_names: Optional[List[str]] = None, worker_kwargs=None, **kwargs):
        ws = worker_names or list(self.socks)
        out = {}
        for i, w in enumerate(ws):
            out[w] = self.request(w, command, **(worker_kwargs[i] if worker_kwargs else kwargs))
        return out

    def get_worker_status(self, worker_name: str) -> WorkerServerStatus:
        try:
            return WorkerServerStatus(name_resolve.get(
                names.worker_status(self.experiment_name, self.trial_name, worker_name)))
        except name_resolve.NameEntryNotFoundError:
            return WorkerServerStatus.UNKNOWN

    def pulse(self) -> Dict[str, WorkerServerStatus]:
        return {w: self.get_worker_status(w) for w in self.socks}

    def close(self):
        for s in self.socks.values():
            try:
                s.close()
            except OSError:
                pass


class Worker:
    """Sub-classes implement `_configure(config) -> WorkerInformation` and `_poll() -> PollResult`."""

    def __init__(self, server: Optional[WorkerServer] = None):
        self._server = server
        self._running = False
        self._exiting = False
        self._configured = False
        self._info: Optional[WorkerInformation] = None
        self.config = None
        self._watch_stop = threading.Event()
        if server is not None:
            server.register_handler("configure", self.configure)
            server.register_handler("reconfigure", self.reconfigure)
            server.register_handler("start", self.start)
            server.register_handler("pause", self.pause)
            server.register_handler("exit", self.exit)
            server.register_handler("interrupt", self.interrupt)
            server.register_handler("ping", lambda: "pong")

    # ----------------------------------------------------------------------- lifecycle
    @property
    def is_configured(self):
        return self._configured

    @property
    def running(self):
        return self._running

    @property
    def exiting(self):
        return self._exiting

    @property
    def worker_index(self) -> int:
        return self._info.worker_index

    @property
    def worker_count(self) -> int:
        return self._info.worker_count

    def _configure(self, config) -> WorkerInformation:
        raise NotImplementedError()

    def _reconfigure(self, **kwargs):
        raise NotImplementedError()

    def _poll(self) -> PollResult:
        raise NotImplementedError()

    def _exit_hook(self, exit_status: WorkerServerStatus):
        pass

    def configure(self, worker_info: WorkerInformation, config=None):
        assert not self._running
        self.config = config
        self.experiment_name, self.trial_name = worker_info.experiment_name, worker_info.trial_name
        self._info = worker_info
        r = self._configure(config)
        if isinstance(r, WorkerInformation):
            self._info = r
        self._configured = True
        if self._info.watch_keys:
            threading.Thread(target=self._watch, args=(list(self._info.watch_keys),), daemon=True).start()
        logger.info(f"Configured successfully: {self._info.worker_name}")

    def reconfigure(self, **kwargs):
        assert not self._running
        self._reconfigure(**kwargs)

    def _watch(self, keys: List[str]):
        """Exit when any watched name_resolve key (a peer's liveness entry) disappears."""
        while not self._watch_stop.is_set() and not self._exiting:
            for k in keys:
                try:
                    name_resolve.get(k)
                except name_resolve.NameEntryNotFoundError:
                    logger.warning(f"watched key {k} is gone: exiting worker {self._info.worker_name}")
                    self.exit()
                    return
            self._watch_stop.wait(WORKER_KEEPALIVE_TTL / 3)

    def start(self):
        self._running = True
        self._set_status(WorkerServerStatus.RUNNING)

    def pause(self):
        self._running = False
        self._set_status(WorkerServerStatus.PAUSED)

    def exit(self, status: WorkerServerStatus = WorkerServerStatus.COMPLETED):
        self._running = False
        self._exiting = True
        self._watch_stop.set()
        self._set_status(status)

    def interrupt(self):
        self.exit(WorkerServerStatus.INTERRUPTED)
        raise KeyboardInterrupt("worker interrupted by the controller")

    def _set_status(self, s: WorkerServerStatus):
        if self._server is not None:
            self._server.set_status(s)

    def _handle_control(self):
        if self._server is not None:
            self._server.handle_requests()

    def run(self):
        self._start_time = time.monotonic()
        logger.debug("Running worker now")
        try:
            while not self._exiting:
                self._handle_control()
                if not self._running:
                    time.sleep(0.05)
                    continue
                r = self._poll()
                if r.sample_count == 0 and r.batch_count == 0:
                    time.sleep(0.002)
            self._exit_hook(WorkerServerStatus.COMPLETED)
        except KeyboardInterrupt:
            self._exit_hook(WorkerServerStatus.INTERRUPTED)
            raise
        except Exception:
            logger.error(f"Worker {getattr(self._info, 'worker_name', '?')} encountered error: "
                         f"{traceback.format_exc()}")
            self._set_status(WorkerServerStatus.ERROR)
            self._exit_hook(WorkerServerStatus.ERROR)
            raise


class AsyncWorker(Worker):
    """Worker whose poll step is a coroutine (`_poll_async`), driven on one private event loop."""

    async def _poll_async(self) -> PollResult:
        raise NotImplementedError()

    def _poll(self) -> PollResult:
        if not hasattr(self, "_loop"):
            self._loop = asyncio.new_event_loop()
            asyncio.set_event_loop(self._loop)
        return self._loop.run_until_complete(self._poll_async())


# worker type name -> "module:Class", resolved lazily by the controller / process entry point
WORKER_TYPES: Dict[str, str] = {
    "master_worker": "areal_vllm_amd.system.master_worker:MasterWorker",
    "model_worker": "areal_vllm_amd.system.model_worker:ModelWorker",
    "generation_server": "areal_vllm_amd.system.generation_server:GenerationServer",
    "gserver_manager": "areal_vllm_amd.system.gserver_manager:GserverManager",
    "rollout_worker": "areal_vllm_amd.system.rollout_worker:RolloutWorker",
}


def load_worker(worker_type: str):
    import importlib

    mod, cls = WORKER_TYPES[worker_type].split(":")
    return getattr(importlib.import_module(mod), cls)


def worker_type_to_module(worker_type: str) -> str:
    return WORKER_TYPES[worker_type].split(":")[0]


def run_worker(worker_type: str, experiment_name: str, trial_name: str, worker_name: str):
    """Process entry point: build the control server and the worker, then loop until told to exit."""
    server = WorkerServer(experiment_name, trial_name, worker_name)
    worker = load_worker(worker_type)(server=server)
    try:
        worker.run()
    finally:
        server.close()
