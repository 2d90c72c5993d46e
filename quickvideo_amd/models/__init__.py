"""Plugin registry — same contract as lvu/models/__init__.py:5-19: every module of this package must expose
`init_lvu_model(model, config)` and `run_lvu_model(self, question, video_path, **gen)`, optionally
`chat_lvu_model(self, messages, **gen)`; the registry key is the file stem."""
import importlib
from pathlib import Path

cur_dir = Path(__file__).parent

lvu_init_model_map = {}
lvu_run_model_map = {}
lvu_chat_model_map = {}

for file in sorted(cur_dir.glob("*.py")):
    if file.name == "__init__.py":
        continue
    module_name = file.stem
    module = importlib.import_module(f".{module_name}", package=__package__)
    assert hasattr(module, "init_lvu_model"), f"Module {module_name} does not have init_lvu_model function."
    assert hasattr(module, "run_lvu_model"), f"Module {module_name} does not have run_lvu_model function."
    lvu_init_model_map[module_name] = module.init_lvu_model
    lvu_run_model_map[module_name] = module.run_lvu_model
    if hasattr(module, "chat_lvu_model"):
        lvu_chat_model_map[module_name] = module.chat_lvu_model

__all__ = list(lvu_init_model_map.keys()) + ["lvu_init_model_map", "lvu_run_model_map", "lvu_chat_model_map"]
