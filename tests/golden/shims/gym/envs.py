import importlib

registry = {}


def register(id, entry_point, kwargs=None, **_):
    registry[id] = (entry_point, dict(kwargs or {}))


def make(id, **kwargs):
    entry_point, base = registry[id]
    if isinstance(entry_point, str):
        mod, fn = entry_point.split(':')
        entry_point = getattr(importlib.import_module(mod), fn)
    kw = dict(base)
    kw.update(kwargs)
    return entry_point(**kw)
