"""HIP side of the package: C-ABI binding (lib), runtime (Act / Tape) and tape-aware operators."""
from . import lib  # noqa: F401
