"""fairseq --user-dir package: registers the DiffNorm model / task / criterion names on import
(fairseq auto-imports `models/` and `tasks/` of a user dir, criterions must be imported by the package:
reference fairseq/utils.py:464-509)."""
from . import registry  # noqa: F401
from . import models, tasks, criterions  # noqa: F401,E402
