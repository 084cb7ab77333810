"""MI355X-native NeRF ray-chunk renderer."""
from . import synthetic  # noqa: F401
