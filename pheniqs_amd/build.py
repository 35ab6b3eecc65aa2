"""Builds libpheniqs_amd.so in-tree with hipcc for gfx950 (pheniqs_amd/csrc/Makefile)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(jobs: int = 8, verbose: bool = False) -> str:
    command = ["make", "-C", os.path.join(HERE, "csrc"), "-j", str(jobs)]
    if not verbose:
        command.append("-s")
    subprocess.check_call(command)
    path = os.path.join(HERE, "libpheniqs_amd.so")
    if not os.path.exists(path):
        raise RuntimeError("build did not produce " + path)
    return path
