"""Plugin registry of the native engine.

Contract kept from the reference (lvu/models/__init__.py:5-19): a plugin is a module of this package that defines
`init_lvu_model(model, config) -> model` and `run_lvu_model(self, question, video_path, **gen)`; `chat_lvu_model(self,
messages, **gen)` is optional; the registry key (= `LVUConfig.model_type`) is the module's name.  The three public
dictionaries carry the reference's names so `lvu.LVU` can bind plugin functions as methods the same way.
"""
import importlib
import pkgutil
from typing import Callable, Dict

REQUIRED = ("init_lvu_model", "run_lvu_model")
OPTIONAL = ("chat_lvu_model",)


def discover() -> Dict[str, Dict[str, Callable]]:
    """Import every sibling module and collect its plugin entry points; a module missing a required one is an error."""
    found: Dict[str, Dict[str, Callable]] = {}
    for info in pkgutil.iter_modules(__path__):
        if info.ispkg or info.name.startswith("_"):
            continue
        mod = importlib.import_module(f"{__name__}.{info.name}")
        missing = [fn for fn in REQUIRED if not callable(getattr(mod, fn, None))]
        if missing:
            raise AssertionError(f"Module {info.name} does not have {missing[0]} function.")
        found[info.name] = {fn: getattr(mod, fn) for fn in REQUIRED + OPTIONAL if callable(getattr(mod, fn, None))}
    return found


_plugins = discover()
lvu_init_model_map = {name: fns["init_lvu_model"] for name, fns in _plugins.items()}
lvu_run_model_map = {name: fns["run_lvu_model"] for name, fns in _plugins.items()}
lvu_chat_model_map = {name: fns["chat_lvu_model"] for name, fns in _plugins.items() if "chat_lvu_model" in fns}

__all__ = sorted(_plugins) + ["lvu_init_model_map", "lvu_run_model_map", "lvu_chat_model_map", "discover"]
