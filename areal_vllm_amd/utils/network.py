"""Host address and free-port helpers (``areal/utils/network.py`` in the reference)."""
import fcntl
import os
import random
import socket
from contextlib import closing
from typing import List, Optional, Set


def gethostname() -> str:
    return socket.gethostname()


def gethostip() -> str:
    try:
        with closing(socket.socket(socket.AF_INET, socket.SOCK_DGRAM)) as s:
            s.connect(("10.255.255.255", 1))
            return s.getsockname()[0]
    except OSError:
        pass
    try:
        return socket.gethostbyname(socket.gethostname())
    except OSError:
        return "127.0.0.1"


def _is_free(port: int) -> bool:
    for fam, host in ((socket.AF_INET, "0.0.0.0"),):
        with closing(socket.socket(fam, socket.SOCK_STREAM)) as s:
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((host, port))
            except OSError:
                return False
    return True


def find_free_ports(count: int, port_range=(10000, 60000), exclude: Optional[Set[int]] = None,
                    lockfile_root: Optional[str] = None) -> List[int]:
    """Pick `count` distinct free TCP ports. With `lockfile_root` (or $AREAL_PORT_LOCKFILE_ROOT) a lock file per
    port keeps concurrent processes on one host from choosing the same port."""
    exclude = set(exclude or ())
    lockfile_root = lockfile_root or os.environ.get("AREAL_PORT_LOCKFILE_ROOT")
    out: List[int] = []
    tries = 0
    while len(out) < count:
        tries += 1
        if tries > 20000:
            raise RuntimeError("could not find enough free ports")
        p = random.randint(*port_range)
        if p in exclude or p in out or not _is_free(p):
            continue
        if lockfile_root:
            os.makedirs(lockfile_root, exist_ok=True)
            path = os.path.join(lockfile_root, f"{p}.lock")
            try:
                fd = os.open(path, os.O_CREAT | os.O_RDWR)
                fcntl.flock(fd, fcntl.LOCK_EX | fcntl.LOCK_NB)
            except OSError:
                continue
        out.append(p)
    return out


def find_multiple_free_ports(count: int, **kw) -> List[int]:
    return find_free_ports(count, **kw)
