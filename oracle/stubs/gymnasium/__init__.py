"""Minimal stand-in for `gymnasium` so that the UNMODIFIED reference under /root/reference
can be imported in the build container (gymnasium/pygame are not installed).

Test infrastructure only: used by oracle/diff_vs_reference.py and tests/golden/make_golden.py.
It implements just the names the reference touches at import/construct time; no arithmetic of
the hot path lives here.
"""
from . import spaces, utils, core
from .core import Env, Wrapper, ActionWrapper, ObservationWrapper
import importlib

_registry = {}


def make(id, **kwargs):
    entry = _registry[id]
    mod, attr = entry.split(":")
    kwargs.pop("max_episode_steps", None)
    return getattr(importlib.import_module(mod), attr)(**kwargs)
