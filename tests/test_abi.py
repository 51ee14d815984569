"""The C-ABI library loads and exports every symbol include/pgrc_match.h declares (no compute calls: no GPU
needed); host-only helpers behave like the reference's parameter derivation."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("pgrc_match.h", "pgrc_mem.h", "pgrc_reads.h"))
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pgrc_(?:match|synth|mem|divider)_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from pgrc_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(_lib.lib, s), f"{s} declared in pgrc_match.h but not exported by libpgrc_match.so"
    assert set(_lib.EXPORTED_SYMBOLS) == set(syms), "python prototypes out of sync with the header"


def test_no_torch_types_in_the_abi_and_no_oracle_linkage():
    import subprocess
    from pgrc_amd import _lib
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "libtorch" not in needed and "libc10" not in needed
    assert "pgrc_oracle" not in needed and "pgrc_ref" not in needed
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "pgrc_or_" not in syms and "pgrc_ref_" not in syms


def test_library_exports_only_the_c_entry_points():
    """pgrc_amd/csrc/exports.map: nothing but unmangled pgrc_* functions is visible to a host that links the library --
    no mangled internals, no kernel stubs or kernel handles"""
    import subprocess
    from pgrc_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert names, "nm printed nothing"
    stray = [n for n in names if not n.startswith("pgrc_")]
    assert not stray, f"exported besides the C entry points: {stray[:8]} ({len(stray)} in all)"
    assert set(header_symbols()) <= set(names)


def test_parameter_derivation_matches_the_reference_rules():
    from pgrc_amd import _lib, copmem_params
    p = _lib.MatchParams()
    assert _lib.lib.pgrc_match_derive_params(100, 38, 50, b"c", C.byref(p)) == 0
    assert (p.read_len, p.seed_len, p.max_mismatches, p.min_mismatches, p.mode) == (100, 38, 2, 0, b"c")
    assert _lib.lib.pgrc_match_derive_params(150, 38, 3, b"C", C.byref(p)) == 0
    assert (p.max_mismatches, p.min_mismatches, p.mode) == (50, 50, b"c")
    assert _lib.lib.pgrc_match_derive_params(100, 100, 50, b"i", C.byref(p)) == 0
    assert p.mode == b"e"  # readLength == seed and mode != c -> DefaultReadsExactMatcher (ReadsMatchers.cpp:715-723)
    assert _lib.lib.pgrc_match_derive_params(100, 400, 50, b"d", C.byref(p)) == 0
    assert (p.seed_len, p.mode) == (100, b"e")
    assert _lib.lib.pgrc_match_derive_params(100, 38, 50, b"q", C.byref(p)) == 7  # "Unknown matching mode"
    assert copmem_params(38, 1_875_000_000) == {"K": 28, "k1": 5, "k2": 2, "hash_size": 1 << 29}
    assert copmem_params(38, 12_500_000)["hash_size"] == 1 << 24
    assert copmem_params(45, 3_100_000_000) == {"K": 32, "k1": 4, "k2": 3, "hash_size": 1 << 30}
    with pytest.raises(Exception):
        copmem_params(20, 1000)


def test_product_fails_loudly_without_a_device():
    """No CPU fallback: without a HIP device every compute entry point reports E_NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pgrc_amd import MatchContext, PgrcMatchError
    with pytest.raises(PgrcMatchError) as e:
        MatchContext(100, 38, 2, 0, "c")
    assert e.value.code == 3


def test_host_generators_are_deterministic_and_paired():
    import numpy as np
    from pgrc_amd import synth
    g = synth.pg_params(100000, seed=9, tandem_every=2)
    a, b = synth.pg_host(g), synth.pg_host(g)
    assert np.array_equal(a, b) and set(np.unique(a)) == set(b"ACGT")
    rs = synth.reads_params(2000, 100, seed=9, paired=True, n_with_n=100)
    r1 = synth.reads_host(g, a, rs)
    r2 = np.concatenate([synth.reads_host(g, a, rs, 0, 700), synth.reads_host(g, a, rs, 700, 1300)])
    assert np.array_equal(r1, r2)                       # shards regenerate the same global read set
    r3 = synth.reads_host(g, None, rs)                  # pure-function path (no text array)
    assert np.array_equal(r1, r3)
    assert (r1[-100:] == ord("N")).any(axis=1).all() and not (r1[:-100] == ord("N")).any()


def test_header_is_plain_c99_and_links_from_c(tmp_path):
    """include/pgrc_match.h compiles as strict C99 and a C program links against the library (what a cgo/FFI
    binding needs); `--dry` touches no compute entry point."""
    import subprocess
    from pgrc_amd import _lib
    exe = tmp_path / "c_abi_smoke"
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Werror", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-o", str(exe), "-L", libdir, "-lpgrc_match",
                        f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), "--dry"], capture_output=True, text=True)
    assert r.returncode == 0 and "dry ok: mode c kmax 2" in r.stdout
