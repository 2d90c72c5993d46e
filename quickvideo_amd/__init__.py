"""quickvideo_amd — MI355X-native QuickPrefill (group-chunked prefill + key-norm KV pruning) behind the
reference's LVU / LVUConfig API.  `from quickvideo_amd import LVU, LVUConfig` mirrors `from lvu import LVU, LVUConfig`."""
from .lvu_config import LVUConfig, LVULayerConfig  # noqa: F401


def __getattr__(name):          # LVU pulls in torch + the plugin registry: import lazily
    if name == "LVU":
        from .lvu import LVU
        return LVU
    raise AttributeError(name)
