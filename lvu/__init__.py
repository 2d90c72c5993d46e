"""Drop-in alias: `from lvu import LVU, LVUConfig` (reference lvu/__init__.py:1-2) resolves to the MI355X-native package."""
from quickvideo_amd.lvu import LVU  # noqa: F401
from quickvideo_amd.lvu_config import LVUConfig, LVULayerConfig  # noqa: F401
