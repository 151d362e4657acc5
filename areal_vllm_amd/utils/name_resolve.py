"""Name resolution: a small KV store used for service discovery, version publication and liveness.

Backends: in-process memory, shared filesystem ("nfs": ``<root>/<name>/ENTRY`` written with
tmp-file + rename), etcd3 (leases + keepalive; needs the ``etcd3`` package) and a Ray actor (needs
``ray``). API and behaviour follow ``areal/utils/name_resolve.py`` in the reference:
``add / add_subentry / delete / clear_subtree / get / get_subtree / find_subtree / wait / watch_names /
reset / reconfigure``; entries added with ``delete_on_exit`` are removed by ``reset`` (and at
interpreter exit), entries with ``keepalive_ttl`` expire unless refreshed.
"""
import atexit
import os
import random
import shutil
import threading
import time
import uuid
from typing import Callable, Dict, List, Optional

from . import logging

logger = logging.getLogger("name-resolve")


class ArgumentError(Exception):
    pass


class NameEntryExistsError(Exception):
    pass


class NameEntryNotFoundError(Exception):
    pass


class NameRecordRepository:
    def __del__(self):
        try:
            self.reset()
        except Exception:
            pass

    def add(self, name, value, delete_on_exit=True, keepalive_ttl=None, replace=False):
        raise NotImplementedError()

    def add_subentry(self, name, value, **kwargs):
        sub_name = name.rstrip("/") + "/" + str(uuid.uuid4())[:8]
        self.add(sub_name, value, **kwargs)
        return sub_name

    def delete(self, name):
        raise NotImplementedError()

    def clear_subtree(self, name_root):
        raise NotImplementedError()

    def get(self, name):
        raise NotImplementedError()

    def get_subtree(self, name_root) -> List[str]:
        raise NotImplementedError()

    def find_subtree(self, name_root) -> List[str]:
        raise NotImplementedError()

    def wait(self, name, timeout=None, poll_frequency=1):
        start = time.monotonic()
        while True:
            try:
                return self.get(name)
            except NameEntryNotFoundError:
                pass
            if timeout is not None and time.monotonic() - start > timeout:
                raise TimeoutError(f"Timeout waiting for key '{name}' ({self.__class__.__name__})")
            time.sleep(min(poll_frequency, 0.05) + random.random() * 0.02)

    def reset(self):
        raise NotImplementedError()

    def watch_names(self, names: List, call_back: Callable, poll_frequency=15, wait_timeout=300):
        """Call `call_back` once when the LAST of `names` disappears (peer liveness watch)."""
        if isinstance(names, str):
            names = [names]
        q = list(names)
        lock = threading.Lock()

        def wrap():
            with lock:
                q.pop()
                if not q:
                    call_back()

        for n in names:
            threading.Thread(target=self._watch_thread_run, args=(n, wrap, poll_frequency, wait_timeout),
                             daemon=True).start()

    def _watch_thread_run(self, name, call_back, poll_frequency, wait_timeout):
        self.wait(name, timeout=wait_timeout, poll_frequency=poll_frequency)
        while True:
            try:
                self.get(name)
                time.sleep(poll_frequency + random.random())
            except NameEntryNotFoundError:
                call_back()
                break


class MemoryNameRecordRepository(NameRecordRepository):
    """Process-local store (unit tests, single-process runs)."""

    def __init__(self, log_events=False):
        self._store: Dict[str, str] = {}
        self._expiry: Dict[str, float] = {}
        self._to_delete = set()
        self._lock = threading.RLock()
        self._log = log_events

    def _expire(self):
        now = time.monotonic()
        for k in [k for k, t in self._expiry.items() if t < now]:
            self._store.pop(k, None)
            self._expiry.pop(k, None)

    def add(self, name, value, delete_on_exit=True, keepalive_ttl=None, replace=False):
        if not name:
            raise ValueError("name cannot be empty")
        name = name.rstrip("/")
        with self._lock:
            self._expire()
            if name in self._store and not replace:
                raise NameEntryExistsError(f"K={name} V={self._store[name]} V2={value}")
            self._store[name] = str(value)
            if keepalive_ttl is not None:
                self._expiry[name] = time.monotonic() + keepalive_ttl
            else:
                self._expiry.pop(name, None)
            if delete_on_exit:
                self._to_delete.add(name)

    def touch(self, name, value, new_time_to_live):
        self.add(name, value, keepalive_ttl=new_time_to_live, replace=True)

    def delete(self, name):
        with self._lock:
            if name not in self._store:
                raise NameEntryNotFoundError(f"K={name}")
            self._to_delete.discard(name)
            self._expiry.pop(name, None)
            del self._store[name]

    def clear_subtree(self, name_root):
        with self._lock:
            root = name_root.rstrip("/")
            for k in [k for k in self._store if k == root or k.startswith(root + "/")]:
                self._store.pop(k)
                self._expiry.pop(k, None)
                self._to_delete.discard(k)

    def get(self, name):
        name = name.rstrip("/")
        with self._lock:
            self._expire()
            if name not in self._store:
                raise NameEntryNotFoundError(f"K={name}")
            return self._store[name]

    def get_subtree(self, name_root):
        with self._lock:
            self._expire()
            root = name_root.rstrip("/")
            return [v for k, v in sorted(self._store.items()) if k == root or k.startswith(root + "/")]

    def find_subtree(self, name_root):
        with self._lock:
            self._expire()
            root = name_root.rstrip("/")
            return sorted(k for k in self._store if k == root or k.startswith(root + "/"))

    def reset(self):
        with self._lock:
            for k in list(self._to_delete):
                self._store.pop(k, None)
                self._expiry.pop(k, None)
            self._to_delete.clear()


class NfsNameRecordRepository(NameRecordRepository):
    """Shared-filesystem store: one directory per key with an ENTRY file; atomic tmp+rename writes.
    TTL entries store their deadline in an ENTRY.ttl side file and are refreshed by a keepalive thread."""

    def __init__(self, record_root="/tmp/areal/name_resolve", **kwargs):
        self.record_root = record_root
        self._to_delete = set()
        self._keepalive: Dict[str, float] = {}
        self._lock = threading.RLock()
        self._ka_thread = None
        self._stop = threading.Event()

    def _dir_path(self, name):
        return os.path.join(self.record_root, name.strip("/"))

    def _file_path(self, name):
        return os.path.join(self._dir_path(name), "ENTRY")

    def _write(self, path, text):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + f".tmp.{uuid.uuid4().hex[:8]}"
        with open(tmp, "w") as f:
            f.write(text)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)

    def add(self, name, value, delete_on_exit=True, keepalive_ttl=None, replace=False):
        if not name:
            raise ValueError("name cannot be empty")
        path = self._file_path(name)
        if self._alive(name) and not replace:
            raise NameEntryExistsError(path)
        self._write(path, str(value))
        if keepalive_ttl is not None:
            self._write(path + ".ttl", repr(time.time() + keepalive_ttl))
            with self._lock:
                self._keepalive[name] = keepalive_ttl
                self._ensure_keepalive()
        elif os.path.exists(path + ".ttl"):
            os.remove(path + ".ttl")
        if delete_on_exit:
            with self._lock:
                self._to_delete.add(name)

    def _ensure_keepalive(self):
        if self._ka_thread is None or not self._ka_thread.is_alive():
            self._stop.clear()
            self._ka_thread = threading.Thread(target=self._keepalive_run, daemon=True)
            self._ka_thread.start()

    def _keepalive_run(self):
        while not self._stop.is_set():
            with self._lock:
                items = list(self._keepalive.items())
            if not items:
                return
            for name, ttl in items:
                path = self._file_path(name)
                if os.path.exists(path):
                    try:
                        self._write(path + ".ttl", repr(time.time() + ttl))
                    except OSError:
                        pass
            self._stop.wait(max(0.2, min(t for _, t in items) / 3))

    def _alive(self, name) -> bool:
        path = self._file_path(name)
        if not os.path.isfile(path):
            return False
        ttl = path + ".ttl"
        if os.path.isfile(ttl):
            try:
                with open(ttl) as f:
                    if float(f.read().strip()) < time.time():
                        return False
            except (OSError, ValueError):
                pass
        return True

    def delete(self, name):
        path = self._file_path(name)
        if not os.path.isfile(path):
            raise NameEntryNotFoundError(path)
        with self._lock:
            self._keepalive.pop(name, None)
            self._to_delete.discard(name)
        for p in (path, path + ".ttl"):
            try:
                os.remove(p)
            except FileNotFoundError:
                pass
        d = os.path.dirname(path)
        while True:  # prune empty parents
            try:
                os.rmdir(d)
            except OSError:
                break
            d = os.path.dirname(d)
            if os.path.abspath(d) == os.path.abspath(self.record_root):
                break

    def clear_subtree(self, name_root):
        d = self._dir_path(name_root)
        with self._lock:
            root = name_root.strip("/")
            for k in [k for k in self._keepalive if k.strip("/").startswith(root)]:
                self._keepalive.pop(k)
        if os.path.isdir(d):
            shutil.rmtree(d, ignore_errors=True)

    def get(self, name):
        path = self._file_path(name)
        for _ in range(20):
            if not self._alive(name):
                raise NameEntryNotFoundError(path)
            try:
                with open(path) as f:
                    return f.read().strip()
            except FileNotFoundError:
                raise NameEntryNotFoundError(path)
            except OSError:
                time.sleep(0.01)
        raise NameEntryNotFoundError(path)

    def find_subtree(self, name_root):
        d = self._dir_path(name_root)
        out = []
        if os.path.isdir(d):
            for root, _, files in os.walk(d):
                if "ENTRY" in files:
                    key = os.path.relpath(root, self.record_root)
                    if self._alive(key):
                        out.append(key)
        return sorted(out)

    def get_subtree(self, name_root):
        vals = []
        for k in self.find_subtree(name_root):
            try:
                vals.append(self.get(k))
            except NameEntryNotFoundError:
                pass
        return vals

    def reset(self):
        self._stop.set()
        with self._lock:
            names = list(self._to_delete)
            self._to_delete.clear()
            self._keepalive.clear()
        for n in names:
            try:
                self.delete(n)
            except (NameEntryNotFoundError, OSError):
                pass


class Etcd3NameRecordRepository(NameRecordRepository):
    """etcd3-backed store with leases for TTL keys (requires the `etcd3` package)."""

    def __init__(self, host=None, port=None, user=None, password=None, **kwargs):
        try:
            import etcd3
        except ImportError as e:
            raise RuntimeError("the etcd3 name-resolve backend needs the `etcd3` package") from e
        self._client = etcd3.client(host=host or "localhost", port=int(port or 2379), user=user, password=password)
        self._lock = threading.RLock()
        self._to_delete = set()
        self._leases = {}
        self._stop = threading.Event()
        self._ka = threading.Thread(target=self._keepalive_run, daemon=True)
        self._ka.start()

    def _keepalive_run(self):
        while not self._stop.wait(1.0):
            with self._lock:
                for lease in list(self._leases.values()):
                    try:
                        lease.refresh()
                    except Exception:
                        pass

    def add(self, name, value, delete_on_exit=True, keepalive_ttl=None, replace=False):
        name = name.rstrip("/")
        with self._lock:
            if not replace and self._client.get(name)[0] is not None:
                raise NameEntryExistsError(name)
            lease = None
            if keepalive_ttl is not None:
                lease = self._client.lease(int(max(1, keepalive_ttl)))
                self._leases[name] = lease
            self._client.put(name, str(value), lease=lease)
            if delete_on_exit:
                self._to_delete.add(name)

    def delete(self, name):
        with self._lock:
            if not self._client.delete(name.rstrip("/")):
                raise NameEntryNotFoundError(name)
            self._to_delete.discard(name)
            self._leases.pop(name, None)

    def clear_subtree(self, name_root):
        with self._lock:
            self._client.delete_prefix(name_root.rstrip("/"))

    def get(self, name):
        v, _ = self._client.get(name.rstrip("/"))
        if v is None:
            raise NameEntryNotFoundError(name)
        return v.decode()

    def get_subtree(self, name_root):
        return [v.decode() for v, _ in sorted(self._client.get_prefix(name_root.rstrip("/")),
                                             key=lambda kv: kv[1].key)]

    def find_subtree(self, name_root):
        return sorted(m.key.decode() for _, m in self._client.get_prefix(name_root.rstrip("/")))

    def reset(self):
        self._stop.set()
        with self._lock:
            for n in list(self._to_delete):
                try:
                    self._client.delete(n)
                except Exception:
                    pass
            self._to_delete.clear()


class RayNameResolveRepository(NameRecordRepository):
    """Store backed by a detached Ray actor (requires `ray`)."""

    def __init__(self, actor_name: str = "ray_kv_store", **kwargs):
        try:
            import ray
        except ImportError as e:
            raise RuntimeError("the ray name-resolve backend needs the `ray` package") from e
        self._ray = ray

        @ray.remote
        class _KV:
            def __init__(self):
                self.repo = MemoryNameRecordRepository()

            def call(self, fn, *a, **k):
                return getattr(self.repo, fn)(*a, **k)

        try:
            self._actor = ray.get_actor(actor_name)
        except ValueError:
            self._actor = _KV.options(name=actor_name, lifetime="detached").remote()
        self._to_delete = set()

    def _call(self, fn, *a, **k):
        return self._ray.get(self._actor.call.remote(fn, *a, **k))

    def add(self, name, value, delete_on_exit=True, keepalive_ttl=None, replace=False):
        self._call("add", name, value, False, keepalive_ttl, replace)
        if delete_on_exit:
            self._to_delete.add(name)

    def delete(self, name):
        self._call("delete", name)
        self._to_delete.discard(name)

    def clear_subtree(self, name_root):
        self._call("clear_subtree", name_root)

    def get(self, name):
        return self._call("get", name)

    def get_subtree(self, name_root):
        return self._call("get_subtree", name_root)

    def find_subtree(self, name_root):
        return self._call("find_subtree", name_root)

    def reset(self):
        for n in list(self._to_delete):
            try:
                self._call("delete", n)
            except Exception:
                pass
        self._to_delete.clear()


def make_repository(args) -> NameRecordRepository:
    t = getattr(args, "type", "nfs")
    if t == "nfs":
        return NfsNameRecordRepository(getattr(args, "nfs_record_root", "/tmp/areal/name_resolve"))
    if t == "memory":
        return MemoryNameRecordRepository()
    if t == "etcd3":
        host, _, port = getattr(args, "etcd3_addr", "localhost:2379").partition(":")
        return Etcd3NameRecordRepository(host=host, port=int(port or 2379))
    if t == "ray":
        return RayNameResolveRepository(actor_name=getattr(args, "ray_actor_name", "ray_kv_store"))
    raise NotImplementedError(f"No such name resolver: {t}")


DEFAULT_REPOSITORY: NameRecordRepository = NfsNameRecordRepository()


def _bind():
    g = globals()
    for fn in ("add", "add_subentry", "delete", "clear_subtree", "get", "get_subtree", "find_subtree", "wait",
               "reset", "watch_names"):
        g[fn] = getattr(DEFAULT_REPOSITORY, fn)


_bind()
atexit.register(lambda: DEFAULT_REPOSITORY.reset())


def reconfigure(config, **kwargs):
    """Switch the module-level repository (called by ``load_expr_config`` on every process). `config` is a
    ``NameResolveConfig``-like object or just the type name with its settings as keywords, e.g.
    ``reconfigure("nfs", nfs_record_root="/tmp/x")`` or ``reconfigure("memory")``."""
    global DEFAULT_REPOSITORY
    if isinstance(config, str):
        import types

        config = types.SimpleNamespace(type=config, **kwargs)
    try:
        DEFAULT_REPOSITORY.reset()
    except Exception:
        pass
    DEFAULT_REPOSITORY = make_repository(config)
    _bind()
