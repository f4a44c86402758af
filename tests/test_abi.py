"""CPU: the C-ABI library builds, loads, and exports exactly what include/fairrec_hip.h declares."""
import ctypes
import os
import re

from fairrec import _C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "fairrec_hip.h")).read()
    return sorted(set(re.findall(r"FR_API\s+[\w\s\*]+?\b(fr_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == _C.exported_names()


def test_library_exports_every_symbol():
    assert os.path.exists(_C.LIB_PATH), "run `python __graft_entry__.py` first"
    h = ctypes.CDLL(_C.LIB_PATH)
    for name in _declared():
        assert hasattr(h, name), name
    lib = _C.lib()
    assert lib.fr_version() >= 1
    # pure host query, no GPU needed
    assert lib.fr_focf_workspace_bytes(8192, 64) > 6 * 8192 * 64 * 4


def test_argument_validation_without_gpu():
    lib = _C.lib()
    rc = lib.fr_sort_segments(None, 8, 10, None, None, None, None, None, None, None)
    assert rc == -1 and b"null" in lib.fr_last_error()
