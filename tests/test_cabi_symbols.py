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


def test_intra_wave_lds_handoffs_go_through_the_fence():
    """VERDICT r02 weak point 9: lanes of a wave hand values to each other through LDS without a workgroup barrier;
    the compiler must be told (a release + acquire fence at wavefront scope around the wave barrier), or it forwards a
    lane's own store past another lane's write.  wave_lds_handoff() is that fence: a bare wave barrier anywhere else
    in the kernels would be such a hand-off without it."""
    csrc = os.path.join(ROOT, "unicorn_amd", "csrc")
    bare = []
    for f in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, f)).read()
        text = re.sub(r"//[^\n]*", "", text)
        # the one allowed use: inside wave_lds_handoff() itself
        body = re.search(r"void wave_lds_handoff\(\)\s*\{(.*?)\n\}", text, flags=re.S)
        if body:
            assert "__ATOMIC_RELEASE" in body.group(1) and "__ATOMIC_ACQUIRE" in body.group(1)
            text = text.replace(body.group(0), "")
        for m in re.finditer(r"__builtin_amdgcn_wave_barrier|__builtin_amdgcn_s_barrier\b", text):
            bare.append((f, text.count("\n", 0, m.start()) + 1))
    assert not bare, f"wave-level barrier outside wave_lds_handoff(): {bare}"
    assert "wave_lds_handoff()" in open(os.path.join(csrc, "ur_render.hip")).read()


def test_library_is_built_with_warnings_as_errors():
    import __graft_entry__ as g
    assert "-Werror" in g.HIP_FLAGS and "-Wall" in g.HIP_FLAGS and "-ffp-contract=off" in g.HIP_FLAGS
