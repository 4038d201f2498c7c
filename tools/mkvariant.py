#!/usr/bin/env python3
"""dev helper: build a variant of the library with textual replacements applied to a copy of ur_render.hip.
usage: mkvariant.py name 'old1=>new1' 'old2=>new2' ...   (each `old` must occur exactly once; @@ stands for a newline)
Output: abx/<name>.hip, abx/lib<name>.so.  Timing experiments only -- a variant may compute wrong numbers."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import HIPCC, HIP_FLAGS, SRC  # noqa: E402


def main():
    name = sys.argv[1]
    src = open(SRC).read()
    for rep in sys.argv[2:]:
        old, new = rep.split("=>", 1)
        old, new = old.replace("@@", "\n"), new.replace("@@", "\n")
        n = src.count(old)
        if n != 1:
            sys.exit(f"pattern occurs {n} times: {old!r}")
        src = src.replace(old, new)
    os.makedirs(os.path.join(ROOT, "abx"), exist_ok=True)
    path = os.path.join(ROOT, "abx", name + ".hip")
    src = src.replace('#include "../../include/unicorn_render.h"', f'#include "{ROOT}/include/unicorn_render.h"')
    src = src.replace('#include "ur_device.h"', f'#include "{ROOT}/unicorn_amd/csrc/ur_device.h"')
    open(path, "w").write(src)
    subprocess.check_call([HIPCC] + HIP_FLAGS + ["-o", os.path.join(ROOT, "abx", f"lib{name}.so"), path])
    print("built", name)


if __name__ == "__main__":
    main()
