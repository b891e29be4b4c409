"""CPU: the C-ABI shared library loads and exports every symbol include/twog_gcn.h declares (no compute calls), the
ctypes struct layouts agree with the C compiler's, and the product refuses to run without the library."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from tests.helpers import ROOT

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import _lib

HEADER = os.path.join(ROOT, 'include', 'twog_gcn.h')


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(twog_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run([sys.executable, '-c', 'import __graft_entry__ as g; g.build()'], cwd=ROOT, check=True)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/twog_gcn.h but not exported'
    assert set(_lib.exported_symbols()) == set(names), set(names) ^ set(_lib.exported_symbols())
    lib.twog_version.restype = ctypes.c_char_p
    assert b'gfx950' in lib.twog_version()


def test_struct_layouts_match_the_c_compiler(tmp_path):
    """sizeof() of every struct as seen by gcc == ctypes.sizeof of its binding."""
    structs = {'twog_rows_t': _lib.Rows, 'twog_gemm_t': _lib.Gemm, 'twog_gru_step_t': _lib.GruStep,
               'twog_gru_step_bwd_t': _lib.GruStepBwd, 'twog_bigru_t': _lib.BiGru, 'twog_bigru_bwd_t': _lib.BiGruBwd,
               'twog_attn_t': _lib.Attn, 'twog_attn_bwd_t': _lib.AttnBwd, 'twog_segrnn_t': _lib.SegRnn,
               'twog_segrnn_bwd_t': _lib.SegRnnBwd, 'twog_gate_t': _lib.Gate, 'twog_loss_t': _lib.Loss,
               'twog_relation_t': _lib.Relation, 'twog_relation_bwd_t': _lib.RelationBwd, 'twog_rowop_t': _lib.RowOp, 'twog_tape_entry_t': _lib.TapeEntry}
    src = tmp_path / 'sz.c'
    body = ''.join(f'printf("{n} %zu\\n", sizeof({n}));' for n in structs)
    src.write_text(f'#include <stdio.h>\n#include "{HEADER}"\nint main(void){{{body}return 0;}}\n')
    exe = tmp_path / 'sz'
    subprocess.run(['gcc', str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    for line in out.strip().splitlines():
        name, size = line.split()
        assert ctypes.sizeof(structs[name]) == int(size), (name, ctypes.sizeof(structs[name]), size)


def test_product_path_fails_loudly_without_the_library(monkeypatch):
    from twog_gcn_amd import kernels
    monkeypatch.setattr(_lib, 'LIB_PATH', os.path.join(ROOT, 'does_not_exist.so'))
    monkeypatch.setattr(_lib, '_lib', None)
    kernels._set_backend_for_tests(None)
    with pytest.raises(RuntimeError, match='no fallback'):
        kernels.get_kernels()
    kernels._set_backend_for_tests(None)
