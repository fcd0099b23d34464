"""Build recipe for the oracle's C restatement (gcc -O2 -fopenmp) -> oracle/_build/liboracle.so.
Test infrastructure; the product never loads it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c", "oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "liboracle.so")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-fno-fast-math", "-ffp-contract=off", "-shared", "-fPIC",
                               SRC, "-o", OUT, "-lm"])
    return OUT


if __name__ == "__main__":
    print(build(force=True))
