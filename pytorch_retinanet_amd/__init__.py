"""MI355X-native RetinaNet dense-head path behind the Python surface of
benihime91/pytorch_retinanet (``from retinanet import Retinanet, AnchorGenerator``,
reference ``retinanet/__init__.py:1-2``)."""
from . import _lib  # noqa: F401  (fails loudly when libretinanet_hip.so is missing)
