"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/eyegaze_hip.h declares
(no compute calls without a GPU), and the ctypes binding lists the same set."""
import re
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (REPO / "include" / "eyegaze_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t|const char\*)\s+(eg_\w+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from eyegaze_multimodal_amd import build
    lib_path = build.build(verbose=False)
    assert lib_path.exists()
    import ctypes
    lib = ctypes.CDLL(str(lib_path))
    names = declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.eg_abi_version.restype = ctypes.c_int
    assert lib.eg_abi_version() == 5


def test_binding_covers_the_header():
    from eyegaze_multimodal_amd import _lib
    names = set(declared_symbols())
    bound = set(_lib.SIGNATURES) | {"eg_abi_version", "eg_last_error", "eg_ffn_gate_bits_bytes"}   # (the last: _lib.gate_bits_bytes)
    assert names <= bound, sorted(names - bound)
    exported = _lib.exported_symbols()
    assert all(exported.get(n, hasattr(_lib.lib(), n)) for n in names), [n for n in names if not exported.get(n)]
    assert _lib.gate_bits_bytes(33280, 1024) == 416 * 8 * 256 * 8      # host-only size helper of eg_ffn_chain's gate bit image


def test_host_side_argument_checks_run_without_a_gpu():
    """Shape/argument validation happens before any launch, so a rejected call is observable on CPU."""
    import ctypes as C
    from eyegaze_multimodal_amd import _lib as L
    d = L.GemmDesc()
    with pytest.raises(L.EgError, match="null operand"):
        L.call("eg_gemm_nt", C.byref(d), 0)
    with pytest.raises(L.EgError, match="bad arguments"):
        L.call("eg_adamw", 0, 0, 0, 0, 10, 0.9, 0.999, 1e-8, 0.01, 0, 0)


def test_layernorm_bwd_rejects_a_grid_larger_than_its_partial_buffer():
    """Round 2's GPU memory access fault: 2080 workgroups each stored a [2, d] partial row into a buffer sized for 2048.  The
    C ABI now takes the buffer's capacity and refuses the call on the host (no launch, so this runs without a GPU); the
    pointers below are never dereferenced."""
    from eyegaze_multimodal_amd import _lib as L
    fake = 0x1000
    with pytest.raises(L.EgError, match="exceeds the partial buffer's capacity of 2048 blocks"):
        L.call("eg_layernorm_bwd", fake, fake, fake, fake, fake, 0, fake, 2080, 2048, 33280, 256, L.EG_BF16, 0.0, 0, 0.0, 0, 0, 0)


def test_engine_refuses_an_out_of_range_ln_block_count(monkeypatch):
    """EYEGAZE_LN_BLOCKS beyond the scratch buffer's 2048 rows used to be clamped silently; now the engine raises."""
    import torch
    from eyegaze_multimodal_amd import DualEEGTransformer
    from eyegaze_multimodal_amd import _lib as L
    from eyegaze_multimodal_amd.engine import Engine
    monkeypatch.setenv("EYEGAZE_LN_BLOCKS", "2080")
    model = DualEEGTransformer(in_channels=8, max_len=256, use_spectrogram=False, use_ibs=False)
    monkeypatch.setattr(Engine, "_alloc", lambda self: None)
    model._flat.ensure(torch.device("cpu"))
    with pytest.raises(L.EgError, match="EYEGAZE_LN_BLOCKS=2080"):
        Engine(model, 4, 1024, torch.device("cpu"), L.EG_BF16)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from eyegaze_multimodal_amd import _lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(L.EgError, match="no CPU fallback"):
        L.lib()
