"""Build id of libkd_engine.so = sha256 over every source the library is compiled from.

One definition, two users: the Makefile bakes `python3 build_id.py` into the library (kd_build_id()),
and `imagen_pytorch/_engine.load()` recomputes it from the sources next to the library (when they are
present) and refuses a binary that was built from other sources.
"""
import hashlib
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent
INCLUDE = CSRC.parent.parent / "include"
SUFFIXES = (".hip", ".h", ".inc")


def source_files():
    files = [p for p in CSRC.iterdir() if p.suffix in SUFFIXES and p.name != "build_id_gen.h"]
    files += [CSRC / "Makefile"]
    files += [p for p in INCLUDE.iterdir() if p.suffix == ".h"]
    return sorted(files, key=lambda p: p.name)


def build_id() -> str:
    h = hashlib.sha256()
    for p in source_files():
        h.update(p.name.encode())
        h.update(b"\0")
        h.update(p.read_bytes())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    sys.stdout.write(build_id())
