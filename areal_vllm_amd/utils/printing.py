"""Tabular rendering of flat statistics dictionaries (role of ``areal/utils/printing.py:6-22``)."""
from itertools import zip_longest
from typing import Dict, List

from tabulate import tabulate


def tabulate_stats(data: Dict[str, float], col: int = 4, floatfmt: str = ".4e") -> str:
    """`col` (name, value) column pairs side by side; the entries run down the first pair, then the second, ..."""
    names: List[str] = list(data)
    height = -(-len(names) // max(1, col))  # ceil
    columns = [names[start:start + height] for start in range(0, len(names), max(1, height))]
    table = []
    for line in zip_longest(*columns):
        row: List[object] = []
        for name in line:
            if name is not None:
                row += [name, data[name]]
        table.append(row)
    return tabulate(table, floatfmt=floatfmt, tablefmt="fancy_grid")
