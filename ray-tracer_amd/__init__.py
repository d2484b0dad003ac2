"""Python glue for the MI355X render path: ctypes bindings of the C ABI (include/rt_amd.h) and of the
host scene front end (ray-tracer_amd/frontend).  PyTorch only provides device memory, streams and
torch.distributed around these calls; no arithmetic of the path happens in Python.

The directory name has a hyphen, so load this package with `rt_loader.load()` (repo root) or through
tests/conftest.py, which registers it as module `ray_tracer_amd`.
"""
from . import binding  # noqa: F401
from . import sharding  # noqa: F401
from .binding import *  # noqa: F401,F403
