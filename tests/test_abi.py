"""CPU: the C-ABI library builds, loads, and exports exactly what include/eavqa.h (the drop-in boundary) and
include/eavqa_test.h (test-only kernel selectors) declare.
No compute call is made here (there is no GPU in the build container)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(headers=("eavqa.h", "eavqa_test.h")):
    out = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(eavqa_[a-z0-9_]+)\s*\(", text))
    return sorted(out)


@pytest.fixture(scope="module")
def lib():
    from eavqa_amd import build, _lib
    build.build()
    return _lib.load()


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    assert "eavqa_gemm" in syms and "eavqa_attention_fwd" in syms and len(syms) >= 20
    public = declared_symbols(("eavqa.h",))
    # the drop-in boundary carries no test hooks and no process-global switches
    assert not [n for n in public if "debug" in n or n.endswith("_ex")]
    assert set(declared_symbols(("eavqa_test.h",))) - set(public) == {"eavqa_gemm_ex", "eavqa_gemm_ln_ex", "eavqa_attention_fwd_ex", "eavqa_attention_bwd_ex",
                                                                                "eavqa_gemm_splitk_ex", "eavqa_lm_block_forward_ex",
                                                                                "eavqa_gemm_decode_ex", "eavqa_t5_decoder_step_ex"}


def test_every_declared_symbol_is_exported_and_bound(lib):
    from eavqa_amd import _lib
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in eavqa.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_the_dynamic_symbol_table_is_exactly_the_two_headers():
    """Hidden visibility + csrc/exports.map: no C++ symbol, kernel handle or toolchain bookkeeping symbol leaves the library."""
    import subprocess
    from eavqa_amd import build
    out = subprocess.run(["nm", "-D", "--defined-only", build.build()], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == declared_symbols()


def test_library_exports_no_mutable_global_switches(lib):
    """`no global mutable state` (include/eavqa.h conventions): the round-1 eavqa_debug_* setters are gone."""
    for name in ("eavqa_debug_disable_fast_gemm", "eavqa_debug_gemm_stagger", "eavqa_debug_attention_valu"):
        assert not hasattr(lib, name), name


def test_abi_version_and_strerror(lib):
    assert lib.eavqa_abi_version() == 1
    assert lib.eavqa_strerror(0) == b"ok"
    assert b"aligned" in lib.eavqa_strerror(-2)
    assert b"unknown" in lib.eavqa_strerror(-99)


def test_argument_validation_happens_before_any_launch(lib):
    """Bad arguments are rejected on the host (no GPU needed): null pointers, bad dtype, bad shapes."""
    assert lib.eavqa_gemm(1, 1, 1, 8, 8, 8, None, 8, None, 8, None, 8, 0, 1.0, None, 0, None, None, 0, None, 0, None) == -1
    assert lib.eavqa_gemm(7, 1, 1, 8, 8, 8, 16, 8, 16, 8, 16, 8, 0, 1.0, None, 0, None, None, 0, None, 0, None) == -4
    assert lib.eavqa_gemm(1, 1, 1, 8, 8, 12, 16, 16, 16, 16, 16, 8, 0, 1.0, None, 0, None, None, 0, None, 0, None) == -3
    assert lib.eavqa_gemm(1, 1, 1, 8, 8, 8, 18, 8, 16, 8, 16, 8, 0, 1.0, None, 0, None, None, 0, None, 0, None) == -2
    assert lib.eavqa_layernorm_fwd(0, 1, 4, 6, 16, 8, None, None, 1e-5, 16, 8, None, None, None) == -3
    assert lib.eavqa_attention_fwd(0, 1, 1, 4, 4, 6, 16, 8, 16, 8, 16, 8, 16, 8, 0, 0, None, 0, None, 0, 1.0, None, None) == -3
    assert lib.eavqa_adamw(0, None, None, None, None, 1, 0.1, 0.9, 0.999, 1e-8, 0.01, 1.0, 0, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from eavqa_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.EavqaError, match="only compute path"):
        _lib.load()


def test_cpu_tensors_are_rejected_not_emulated():
    import torch
    from eavqa_amd import ops, _lib
    with pytest.raises(_lib.EavqaError, match="no CPU fallback"):
        ops.gemm(torch.zeros(8, 8), torch.zeros(8, 8))


def test_decode_plans_are_pure_host_arithmetic(lib):
    """eavqa_gemm_splitk_plan / eavqa_lm_block_workspace_bytes touch no device: the split of every decode GEMM of the few-shot model
    (OPT-2.7B, B = 32) is the one profiles/round2_decode.md was measured with (FFN-up since round 3: 512-deep slices, no split gives one
    workgroup per CU there), unsupported shapes give 0."""
    plan = lib.eavqa_gemm_splitk_plan
    E, F = 2560, 10240
    assert [plan(32, 3 * E, E), plan(32, E, E), plan(32, F, E), plan(32, E, F)] == [4, 10, 5, 10]
    for M, N, K in ((32, 3 * E, E), (1, 64, 32), (64, 50272, 2560), (17, 200, 96), (64, 12800, 512)):
        ks = plan(M, N, K)
        assert ks >= 1 and K % (32 * ks) == 0 and (K // ks) * (1 if M <= 16 else 2 if M <= 32 else 4) * 32 <= 64 * 1024
    assert plan(32, 8192, 2048) == 4 and plan(32, 16384, 4096) == 8        # OPT-1.3B FFN-up: one round exists; OPT-6.7B: 512-deep slices
    assert plan(65, 128, 64) == 0 and plan(8, 128, 48) == 0 and plan(0, 128, 64) == 0
    ws = lib.eavqa_lm_block_workspace_bytes
    small, big = ws(1, 32, E, F), ws(1, 4800, E, F)             # dtype 1 = bfloat16: a decode step, the 150-position prefill of 32 prompts
    assert 0 < small < big
    assert 0 < ws(0, 32, E, F) < small            # fp32 has no split-K decode route: no partial-sum buffers in its workspace
