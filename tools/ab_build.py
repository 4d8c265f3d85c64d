#!/usr/bin/env python
"""Build a named variant of libesvo_hip.so for A/B measurements on the GPU box (the .so travels with the snapshot).
usage: python tools/ab_build.py <name> [--rev <git rev> file ... | --rev <git rev> ALL] [-D...flags]
  --rev R f1 f2 : take these source files (relative to esvo_amd/csrc) from git revision R instead of the working tree
  --rev R ALL   : the whole library (esvo_amd/csrc + include) as of revision R
writes tools/ab/libesvo_hip_<name>.so; run with ESVO_HIP_LIB=tools/ab/libesvo_hip_<name>.so"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from esvo_amd import lib  # noqa: E402


def main():
    name = sys.argv[1]
    args = sys.argv[2:]
    rev, files, flags = None, [], []
    i = 0
    while i < len(args):
        if args[i] == "--rev":
            rev = args[i + 1]
            i += 2
            while i < len(args) and not args[i].startswith("-"):
                files.append(args[i])
                i += 1
        else:
            flags.append(args[i])
            i += 1
    csrc = os.path.join(ROOT, "esvo_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "esvo_amd", "csrc")   # common.hpp includes ../../include/esvo_hip.h
        shutil.copytree(csrc, src, ignore=shutil.ignore_patterns("*.so"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
        shutil.copytree(os.path.join(ROOT, "tools", "dev_hooks"), os.path.join(d, "tools", "dev_hooks"))   # (dev_hooks.hpp includes ../../tools/dev_hooks/ under -D flags)
        if files == ["ALL"]:
            for sub in ("esvo_amd/csrc", "include"):
                shutil.rmtree(os.path.join(d, sub))
                os.makedirs(os.path.join(d, sub))
                names = subprocess.check_output(["git", "-C", ROOT, "ls-tree", "--name-only", f"{rev}:{sub}"]).decode().split()
                for f in names:
                    blob = subprocess.check_output(["git", "-C", ROOT, "show", f"{rev}:{sub}/{f}"])
                    open(os.path.join(d, sub, f), "wb").write(blob)
            files = []
        for f in files:
            blob = subprocess.check_output(["git", "-C", ROOT, "show", f"{rev}:esvo_amd/csrc/{f}"])
            open(os.path.join(src, f), "wb").write(blob)
        out_dir = os.path.join(ROOT, "tools", "ab")
        os.makedirs(out_dir, exist_ok=True)
        out = os.path.join(out_dir, f"libesvo_hip_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc"] + lib.HIPCC_FLAGS + flags + ["-I", os.path.join(d, "include"), "-o", out] + \
              [os.path.join(src, s) for s in lib._SOURCES if os.path.exists(os.path.join(src, s))]
        subprocess.check_call(cmd)
        print(out)


if __name__ == "__main__":
    main()
