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


# The reference's own registry keys (lvu/models/: qwen25_lvu_interleaved.py = overlapped producer, qwen25_lvu.py = fetch-then-prefill,
# qwen25_vl.py = the older copy of the latter and LVUConfig's default there) resolve to the native plugin with the same schedule, so a
# reference user's `LVUConfig(..., model_type="qwen25_lvu_interleaved")` runs unchanged.
REFERENCE_ALIASES = {"qwen25_lvu_interleaved": "qwen2vl_mi355x", "qwen25_lvu": "qwen2vl_mi355x_sequential", "qwen25_vl": "qwen2vl_mi355x_sequential"}

_plugins = discover()
for _alias, _target in REFERENCE_ALIASES.items():
    _plugins.setdefault(_alias, _plugins[_target])
lvu_init_model_map = {name: fns["init_lvu_model"] for name, fns in _plugins.items()}
lvu_run_model_map = {name: fns["run_lvu_model"] for name, fns in _plugins.items()}
lvu_chat_model_map = {name: fns["chat_lvu_model"] for name, fns in _plugins.items() if "chat_lvu_model" in fns}

__all__ = sorted(n for n in _plugins if n not in REFERENCE_ALIASES) + ["lvu_init_model_map", "lvu_run_model_map", "lvu_chat_model_map", "discover",
                                                                          "REFERENCE_ALIASES"]
