"""Tabular rendering of flat statistics dictionaries (``areal/utils/printing.py:6-22``)."""
from typing import Dict

from tabulate import tabulate


def tabulate_stats(data: Dict[str, float], col: int = 4, floatfmt: str = ".4e") -> str:
    """Lay the (key, value) pairs out in `col` column pairs, filled column by column."""
    items = list(data.items())
    rows = (len(items) + col - 1) // col
    table = [[x for j in range(col) if i + j * rows < len(items) for x in items[i + j * rows]] for i in range(rows)]
    return tabulate(table, floatfmt=floatfmt, tablefmt="fancy_grid")
