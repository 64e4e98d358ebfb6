"""Compile the C part of the oracle (TEST INFRASTRUCTURE ONLY) with gcc into oracle/liboracle.so."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "pisco_oracle.c")
LIB = os.path.join(HERE, "liboracle.so")


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-o", LIB, SRC])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
