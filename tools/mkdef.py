#!/usr/bin/env python3
"""dev helper: build the library with extra -D switches.  usage: mkdef.py name [-DX=1 ...]  -> abx/lib<name>.so"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import HIPCC, HIP_FLAGS, SRC  # noqa: E402

os.makedirs(os.path.join(ROOT, "abx"), exist_ok=True)
out = os.path.join(ROOT, "abx", f"lib{sys.argv[1]}.so")
subprocess.check_call([HIPCC] + HIP_FLAGS + sys.argv[2:] + ["-o", out, SRC])
print("built", out)
