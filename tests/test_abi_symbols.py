"""The C-ABI library loads and exports every symbol include/hbmpc_hip.h declares (no GPU needed:
nothing is called), and refuses to work without a device."""
import ctypes as C
import os
import re

import pytest

from __graft_entry__ import PKG_DIR, ROOT, load_package


def _declared():
    txt = open(os.path.join(ROOT, "include", "hbmpc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hbmpc_[a-z0-9_]+)\s*\(", txt)))


def test_exports_every_declared_symbol():
    pkg = load_package()
    lib = pkg.lib()
    names = _declared()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.hbmpc_version().startswith(b"hbmpc-hip")


def test_library_is_in_tree():
    pkg = load_package()
    assert os.path.dirname(pkg.hbmpc.LIB_PATH) == PKG_DIR and os.path.exists(pkg.hbmpc.LIB_PATH)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="has a GPU")
def test_no_cpu_path_without_gpu():
    pkg = load_package()
    with pytest.raises(pkg.HbmpcError):
        pkg.Engine(0)
    ctx = C.c_void_p()
    assert pkg.lib().hbmpc_create(C.c_int(-1), C.c_int(0), C.byref(ctx)) == 100  # HBMPC_NO_DEVICE
    assert not ctx


def test_product_never_touches_the_oracle():
    # the product tree must not import, link or load anything under oracle/
    bad = []
    for dirpath, _, files in os.walk(PKG_DIR):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".inc", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle|libhbmpc_oracle|hbmpc_oracle\.h|#include\s+\"[^\"]*oracle|oracle/.*\.(c|so)",
                             txt, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
