"""Per-user scratch directory (``areal/utils/fs.py:1-9``)."""
import getpass
import os


def get_user_tmp() -> str:
    for base in (os.path.join("/home", getpass.getuser(), ".cache"), "/tmp"):
        d = os.path.join(base, "areal")
        try:
            os.makedirs(d, exist_ok=True)
            return d
        except OSError:
            continue
    raise OSError("no writable scratch directory")
