"""Named, optionally coloured loggers (role of ``areal/utils/logging.py`` in the reference)."""
import logging as _logging
import os
import sys
from typing import Optional

_COLORS = {"DEBUG": "\033[36m", "INFO": "\033[32m", "WARNING": "\033[33m", "ERROR": "\033[31m",
           "CRITICAL": "\033[35m"}
_RESET = "\033[0m"
_DATE = "%Y%m%d-%H:%M:%S"
_FMT = "%(asctime)s.%(msecs)03d %(name)s %(levelname)s: %(message)s"
_configured = set()


class _ColorFormatter(_logging.Formatter):
    def format(self, record):
        msg = super().format(record)
        if sys.stderr.isatty() and os.environ.get("NO_COLOR") is None:
            return f"{_COLORS.get(record.levelname, '')}{msg}{_RESET}"
        return msg


def getLogger(name: Optional[str] = None, type_: Optional[str] = None) -> _logging.Logger:
    name = name or "areal"
    logger = _logging.getLogger(name)
    if name not in _configured:
        h = _logging.StreamHandler(sys.stderr)
        h.setFormatter(_ColorFormatter(_FMT, _DATE))
        logger.addHandler(h)
        logger.propagate = False
        logger.setLevel(os.environ.get("AREAL_LOG_LEVEL", "INFO").upper())
        _configured.add(name)
    return logger


def add_file_handler(logger: _logging.Logger, path: str):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    h = _logging.FileHandler(path)
    h.setFormatter(_logging.Formatter(_FMT, _DATE))
    logger.addHandler(h)


DEBUG, INFO, WARNING, ERROR = _logging.DEBUG, _logging.INFO, _logging.WARNING, _logging.ERROR
