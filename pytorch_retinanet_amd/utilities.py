"""``ifnone`` -- the None-coalescing helper used by every constructor
(reference ``retinanet/utilities.py:4-10``)."""
from typing import Any


def ifnone(a: Any, b: Any) -> Any:
    return b if a is None else a
