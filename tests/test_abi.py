"""CPU: the C-ABI library loads and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        src = open(h).read()
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        names |= set(re.findall(r'\b(ifh_[a-z0-9_]+)\s*\(', src))
    return names


def test_every_declared_symbol_is_exported(built_lib):
    names = declared_symbols()
    assert len(names) >= 15
    missing = [n for n in sorted(names) if not hasattr(built_lib, n)]
    assert not missing, missing


def test_binding_table_covers_header():
    from infernos_amd import _lib
    assert declared_symbols() == set(_lib.SIGNATURES), declared_symbols() ^ set(_lib.SIGNATURES)


def test_host_only_entry_points(built_lib, golden_dir):
    import numpy as np
    assert built_lib.ifh_version() >= 100
    a = np.zeros(256, np.int16)
    b = np.zeros(65536, np.uint8)
    assert built_lib.ifh_g711_tables_host(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p)) == 0
    g = np.load(os.path.join(golden_dir, 'g711_tables.npz'))
    assert np.array_equal(a, g['ulaw_to_pcm']) and np.array_equal(b, g['pcm_to_ulaw'])
    assert built_lib.ifh_g711_tables_host(None, None) < 0
    assert b'out256_host' in built_lib.ifh_last_error()


def test_no_oracle_import_in_product():
    """The product never imports the oracle (or any CPU fallback module)."""
    for path in glob.glob(os.path.join(ROOT, 'infernos_amd', '**', '*.py'), recursive=True):
        src = open(path).read()
        assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), path
    for path in glob.glob(os.path.join(ROOT, 'infernos_amd', 'csrc', '*')):
        assert 'oracle/' not in open(path).read() or path.endswith('dsp.hip') and 'oracle/dsp_oracle.c' in open(path).read()


def test_compute_without_device_fails_loudly():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip('has a device')
    from infernos_amd import _lib
    from infernos_amd.codecs import G711Codec
    with pytest.raises(_lib.InfernosHipError):
        G711Codec().decode(b'\xff' * 160)
