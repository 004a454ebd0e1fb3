"""Import shim: makes the hyphenated directory `mobilenet-yolo-pytorch_amd/` importable as the
package `mobilenet_yolo_pytorch_amd` (a module with `__path__` is a package)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mobilenet-yolo-pytorch_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
