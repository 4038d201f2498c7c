"""CPU-only: the C-ABI library loads and exports every symbol include/unicorn_render.h declares (no compute calls)."""
import os
import re

from unicorn_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "unicorn_render.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ur_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _header_symbols() == native.exported_symbols()


def test_library_exports_every_declared_symbol():
    handle = native.lib()
    for name in _header_symbols():
        assert hasattr(handle, name), name
    assert handle.ur_abi_version() >= 1
    assert handle.ur_workspace_bytes(10, 1, 8, 8, 1) >= 10 * 96
    assert handle.ur_last_error() is not None


def test_product_package_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "unicorn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/", "oracle/").lower() or f == "synth.py" and False, \
                    f"{f} mentions the oracle"
