"""Surface selected methods of one object on another.

A class marks the methods it offers with ``@wrapable()`` (optionally under another name); ``wrap(target,
source)`` then gives `target` an attribute per marked method of `source`. With a `transform`, the attribute
is the transform bound to that method: it is called as ``transform(*args, wrap_method_name=...,
wrap_original_method=..., **kwargs)`` and uses the ``wrap_get_*`` / ``wrap_remove_meta`` accessors to take
its two extra keyword arguments apart (a controller-side proxy forwards the call to a remote engine this
way). Contract of ``areal/utils/wrapper.py``.
"""
from typing import Any, Callable, Dict, Iterator, Optional, Tuple

_TAG = "__wrap_meta__"
_NAME_KEY, _METHOD_KEY = "wrap_method_name", "wrap_original_method"


def wrapable(name: Optional[str] = None):
    def tag(fn):
        setattr(fn, _TAG, {"name": name if name else fn.__name__})
        return fn

    return tag


def _offered(source: object) -> Iterator[Tuple[str, Callable]]:
    for attr in dir(source):
        try:
            member = getattr(source, attr)
        except AttributeError:  # properties that need state the object does not have yet
            continue
        info = getattr(member, _TAG, None)
        if callable(member) and info is not None:
            yield info["name"], member


def _forwarder(transform: Callable, public_name: str, method: Callable) -> Callable:
    def call(*args: Any, **kwargs: Any):
        return transform(*args, **{_NAME_KEY: public_name, _METHOD_KEY: method}, **kwargs)

    call.__name__ = public_name
    call.__doc__ = method.__doc__
    return call


def wrap(target: object, source: object, transform: Optional[Callable] = None) -> None:
    for public_name, method in _offered(source):
        setattr(target, public_name, method if transform is None else _forwarder(transform, public_name, method))


def wrap_get_method_name(kwargs: Dict) -> str:
    return kwargs[_NAME_KEY]


def wrap_get_method(kwargs: Dict) -> Callable:
    return kwargs[_METHOD_KEY]


def wrap_remove_meta(kwargs: Dict) -> Dict:
    for k in (_NAME_KEY, _METHOD_KEY):
        kwargs.pop(k, None)
    return kwargs
