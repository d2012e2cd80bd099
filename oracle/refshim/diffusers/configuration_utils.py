import functools
import inspect
from types import SimpleNamespace


class ConfigMixin:
    config_name = "config.json"

    def register_to_config(self, **kwargs):
        if not hasattr(self, "config"):
            self.config = _CfgHolder()
        for k, v in kwargs.items():
            setattr(self.config, k, v)


class _CfgHolder(SimpleNamespace):
    pass


class _Cfg(SimpleNamespace):
    def __getitem__(self, k):
        return getattr(self, k)

    def get(self, k, d=None):
        return getattr(self, k, d)


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        init(self, *args, **kwargs)
        self.config = _Cfg(**cfg)

    return inner
