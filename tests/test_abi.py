"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/mnyolo.h declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    import mobilenet_yolo_pytorch_amd.build as b
    return b.build()


def test_one_translation_unit_really_goes_through_hipcc(tmp_path):
    """build() is mtime-incremental: on a box that receives prebuilt objects it only links.  Force the smallest source through the
    compiler with the build's own flags so that "the sources compile for gfx950" is exercised whatever the state of _obj/."""
    import subprocess
    import mobilenet_yolo_pytorch_amd.build as b
    src = os.path.join(b.CSRC, "core.hip")
    out = str(tmp_path / "core.o")
    r = subprocess.run([b.HIPCC] + b.BASE_FLAGS + b.EXTRA.get("core.hip", []) + ["-c", src, "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "warning" not in r.stderr, r.stderr[-2000:]
    assert os.path.getsize(out) > 1000
    sym = subprocess.run(["nm", out], capture_output=True, text=True).stdout
    assert " T mny_version" in sym and " T mny_last_error" in sym


def _declared():
    src = open(os.path.join(REPO, "include", "mnyolo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mny_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "missing export " + n


def test_binding_table_matches_header(lib_path):
    from mobilenet_yolo_pytorch_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()
    assert _lib.load().mny_version() == 100


def test_argument_errors_do_not_need_a_gpu(lib_path):
    from mobilenet_yolo_pytorch_amd import _lib
    with pytest.raises(_lib.MnyError, match="null pointer"):
        _lib.call("mny_dw_fwd", None, None, None, 0, None, None, None, 1, 8, 8, 32, 3, 1, None)
    assert _lib.load().mny_dw_stat_parts(1, 8, 8, 30, 3, 1) < 0          # C not a multiple of 4
    assert _lib.load().mny_pw_stat_parts(1000, 32, 64) > 0


def test_product_has_no_oracle_import():
    pkg = os.path.join(REPO, "mobilenet-yolo-pytorch_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
