"""TEST INFRASTRUCTURE ONLY -- a minimal stand-in for the `fairseq` package: just the registry decorators with the
checks the DiffNorm fork's decorators perform (duplicate name, duplicate class name, base class), restated from their
behaviour (reference fairseq/models/__init__.py:109-207, fairseq/tasks/__init__.py:48-101, fairseq/registry.py:62-100),
plus `_fork.py`, which pre-registers the names and class names the fork itself registers.  It lets the CPU suite drive
diffnorm_amd/fairseq_plugin/registry.py down its fairseq-present branch (tests/test_plugin_registry.py)."""
from . import criterions, models, tasks  # noqa: F401
from . import _fork  # noqa: F401,E402
