"""Re-export marked methods of one object on another (optionally through a transform).

Used to surface engine methods on controller-side proxies: methods decorated with ``@wrapable()`` on the
source are attached to the target under their (possibly renamed) name; with a `transform`, the target
method becomes ``transform(*args, wrap_method_name=..., wrap_original_method=..., **kwargs)``.
Contract of ``areal/utils/wrapper.py:1-47``.
"""
import functools
import inspect
from typing import Callable, Dict, Optional

_META = "__wrap_meta__"


def wrapable(name: Optional[str] = None):
    def mark(fn):
        setattr(fn, _META, {"name": name or fn.__name__})
        return fn

    return mark


def wrap(target: object, source: object, transform: Optional[Callable] = None) -> None:
    for _, member in inspect.getmembers(source, predicate=callable):
        meta = getattr(member, _META, None)
        if meta is None:
            continue
        exported = member if transform is None else functools.partial(
            transform, wrap_method_name=meta["name"], wrap_original_method=member)
        setattr(target, meta["name"], exported)


def wrap_get_method_name(kwargs: Dict) -> str:
    return kwargs["wrap_method_name"]


def wrap_get_method(kwargs: Dict) -> Callable:
    return kwargs["wrap_original_method"]


def wrap_remove_meta(kwargs: Dict) -> Dict:
    kwargs.pop("wrap_method_name", None)
    kwargs.pop("wrap_original_method", None)
    return kwargs
