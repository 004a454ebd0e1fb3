"""mobilenet-yolo-pytorch_amd — MI355X-native MobileNet-YOLO hot path.

Import as `mobilenet_yolo_pytorch_amd` (the repo-root shim maps the hyphenated directory)."""
from . import _lib  # noqa: F401
from ._lib import MnyError  # noqa: F401
from .model import yolo, HeadState  # noqa: F401
from . import mbv3  # noqa: F401
