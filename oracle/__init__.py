"""ORACLE — test infrastructure only (see oracle/ops.py header).  PARITY UNPINNED."""
from . import ops, net  # noqa: F401
