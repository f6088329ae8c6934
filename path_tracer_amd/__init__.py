"""path_tracer_amd — MI355X-native renderer for path_tracer's per-pixel ray loop.

The rendering itself lives in ``libpt_hip.so`` (hand-written HIP for gfx950
behind the C-ABI of ``include/pt_hip.h``); this package is the thin host layer:
record layouts (``types``), scene builders (``scenes``), and ``HipRenderer``
(the reference's ``VulkanRenderer`` interface).
"""
from . import scenes, types  # noqa: F401
from .renderer import HipRenderer, render_scene  # noqa: F401

__all__ = ["HipRenderer", "render_scene", "scenes", "types"]
