"""CPU-side checks of the boundary: the shared library loads without a GPU and exports every entry
point that include/piscoseg.h declares; no compute call is made here."""
import ctypes
import os
import re

import pytest

from imageprocessing_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "piscoseg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pseg_[a-z_0-9]+)\s*\(", text)))


def test_header_and_loader_agree():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name


def test_table_struct_matches_header():
    # field order of the ctypes mirror follows the C struct
    text = open(os.path.join(ROOT, "include", "piscoseg.h")).read()
    body = text[text.index("typedef struct pseg_table {") + len("typedef struct pseg_table {"):text.index("} pseg_table;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or decl.startswith("typedef"):
            continue
        for part in decl.split(","):
            m = re.search(r"\*?\s*([A-Za-z_0-9]+)\s*$", part.strip())
            names.append(m.group(1))
    assert names == [f[0] for f in _lib.Table._fields_]


def test_create_fails_cleanly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from imageprocessing_amd.engine import PsegError, Segmenter
    with pytest.raises(PsegError):
        Segmenter(64, 64, max_frames=4)
