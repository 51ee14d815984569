"""End-to-end drop-in: the reference's whole encoder with HipReadsMatcher in its matcher seam must write the same
archive, byte for byte, as the untouched reference at -t 1, and that archive must decode to the input reads
(tests/e2e_dropin.py does the work in a child process; oracle/Makefile explains the mapReadsIntoPg interposition)."""
import json
import os
import subprocess
import sys

import pytest

import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def _have_e2e():
    return orc.have_ref() and hasattr(orc.ref(), "pgrc_ref_encode")


def _run(tmp_path, case, cpu_only, devices=None, extra_env=None):
    env = dict(os.environ)
    env.pop("PGRC_REF_VERBOSE", None)
    env.pop("PGRC_DEVICES", None)
    if devices:
        env["PGRC_DEVICES"] = devices
    env.update(extra_env or {})
    if cpu_only:
        env["PGRC_E2E_CPU_ONLY"] = "1"
    p = subprocess.run([sys.executable, os.path.join(HERE, "e2e_dropin.py"), str(tmp_path), case], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, f"e2e child failed ({p.returncode}):\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}"
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("case", ["se", "pe_order"])
def test_reference_encoder_harness_cpu(tmp_path, case):
    """The harness itself, no GPU: the compiled reference encoder is deterministic at -t 1 and round-trips."""
    if not _have_e2e():
        pytest.skip("oracle/_ref was built without the encoder harness")
    r = _run(tmp_path, case, cpu_only=True)
    assert r["identical"] and r["roundtrip"], r
    assert r["cpu_gpu_calls"] == 0 and r["gpu_gpu_calls"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["se", "se_order", "pe", "pe_order", "se_pre", "se_modeD", "se_modeI", "se_exact"])
def test_archive_identical_with_gpu_matcher(tmp_path, case):
    if not _have_e2e():
        pytest.skip("oracle/_ref was built without the encoder harness")
    r = _run(tmp_path, case, cpu_only=False)
    assert r["cpu_gpu_calls"] == 0 and r["gpu_gpu_calls"] >= 1, r     # the GPU leg really went through HipReadsMatcher
    assert r["cpu_text_match_calls"] == 0 and r["gpu_text_match_calls"] >= 3, r   # ... and HipTextMatcher (lq, n, hq Pg)
    assert r["identical"], r
    assert r["roundtrip"], r
    if case not in ("se_modeD", "se_modeI", "se_exact"):     # (those match nothing: the sum-set quirk)
        assert r["gpu_device_exports"] >= 1 and r["cpu_device_exports"] == 0, r   # the export streams came from the device
    if case in ("se", "se_order", "pe", "pe_order"):
        assert r["gpu_streamed_runs"] >= 1, r               # stage 4 took the pipelined hand-over (first phase of mode c)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["se", "pe_order"])
def test_archive_identical_with_the_steps_in_turn(tmp_path, case):
    """PGRC_NO_STREAM=1: hand-over, run and result fetch one after the other (the path every other configuration takes)."""
    if not _have_e2e():
        pytest.skip("oracle/_ref was built without the encoder harness")
    r = _run(tmp_path, case, cpu_only=False, extra_env={"PGRC_NO_STREAM": "1"})
    assert r["gpu_gpu_calls"] >= 1 and r["gpu_streamed_runs"] == 0 and r["identical"] and r["roundtrip"], r


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["se", "pe_order"])
def test_archive_identical_with_a_matcher_over_two_logical_devices(tmp_path, case):
    """PGRC_DEVICES=0,0: HipReadsMatcher builds ONE matcher over two shards (pgrc_match_create_multi; on a box with
    two GPUs: PGRC_DEVICES=0,1 and a RCCL all-gather).  Same archive, byte for byte; the export streams come from the
    first device, where the library gathers the shards' results and reads."""
    if not _have_e2e():
        pytest.skip("oracle/_ref was built without the encoder harness")
    r = _run(tmp_path, case, cpu_only=False, devices="0,0")
    assert r["gpu_gpu_calls"] >= 1 and r["identical"] and r["roundtrip"], r
    assert r["gpu_device_exports"] >= 1, r


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["se", "pe_order"])
def test_archive_identical_with_the_screened_schedule_forced(tmp_path, case):
    """PGRC_SCREEN=1: the two-pass runs of the encoder take the screened schedule (exact-match screen on the RC text first)
    although its 100-bp reads are below the length where the library picks it by itself.  Same archive, byte for byte."""
    if not _have_e2e():
        pytest.skip("oracle/_ref was built without the encoder harness")
    r = _run(tmp_path, case, cpu_only=False, extra_env={"PGRC_SCREEN": "1"})
    assert r["gpu_gpu_calls"] >= 1 and r["identical"] and r["roundtrip"], r


@pytest.mark.gpu
@pytest.mark.parametrize("case,read_len,env", [("se", 100, {"PGRC_DUAL": "1"}), ("pe_order", 100, {"PGRC_DUAL": "1"}),
                                               ("se", 150, {}), ("pe", 150, {}), ("se_order", 150, {})])
def test_archive_identical_with_the_dual_kernel(tmp_path, case, read_len, env):
    """The encoder's stage 4 through the dual kernel (one query per read over both strands): forced on 100-bp reads
    (PGRC_DUAL=1; 37 seeds per read are below the length where the library picks it), and on 150-bp reads, where it is
    the library's own choice -- at PgRC's shipped -M 3, i.e. k <= 33 / k <= 50.  Same archive, byte for byte."""
    if not _have_e2e():
        pytest.skip("oracle/_ref was built without the encoder harness")
    r = _run(tmp_path, case, cpu_only=False, extra_env=dict(env, PGRC_E2E_READ_LEN=str(read_len)))
    assert r["gpu_gpu_calls"] >= 1 and r["gpu_dual_runs"] >= 1, r
    assert r["identical"] and r["roundtrip"], r


@pytest.mark.gpu
@pytest.mark.parametrize("case,env", [("se", {}), ("pe_order", {}), ("pe", {"PGRC_DIVIDE_FROM_ROWS": "1"}),
                                      ("se", {"PGRC_REF_Q_PROMILS": "50"}),                              # one-position quality test
                                      ("se", {"PGRC_REF_Q_PROMILS": "120", "PGRC_REF_Q_FULL": "1"}),     # arithmetic-mean test
                                      ("pe", {"PGRC_REF_Q_PROMILS": "200", "PGRC_REF_Q_FULL": "1"})])
# (N reads in the LQ set -- PgRC's -N -- together with a division ends in the reference's "Unimplemented transferring reads between
#  reads sets packed with different alphabet": that combination is covered at the factory level, tests/test_gpu_divide.py)
def test_archive_identical_with_the_read_sets_made_on_the_device(tmp_path, case, env):
    """Row f3 inside the whole encoder: DividedPCLReadsSets' factories replaced by integration/HipDividedReadsSets -- the FASTQ
    files go to the device as text, in pieces of 100 000 bytes, are parsed there (lines, records, the pair file's reads reverse-
    complemented), and classification, packing and the LQ / N mappings come back; PGRC_DIVIDE_FROM_ROWS: the records are taken
    from the reference's iterator instead, 7 000 per batch -- on top of the GPU matcher and text matcher.  Same archive as the
    untouched reference, byte for byte, also with PgRC's quality-based division in both of its forms."""
    if not _have_e2e() or not hasattr(orc.ref(), "pgrc_ref_division_calls"):
        pytest.skip("oracle/_ref was built without the division harness")
    r = _run(tmp_path, case, cpu_only=False, extra_env=dict(env, PGRC_E2E_GPU_STAGES="7", PGRC_DIVIDE_BATCH="7000", PGRC_FASTQ_PIECE="100000"))
    assert r["gpu_division_calls"] >= 1 and r["gpu_gpu_calls"] >= 1, r
    assert r["identical"] and r["roundtrip"], r


@pytest.mark.gpu
@pytest.mark.parametrize("case,threads", [("se", "1"), ("pe", "1"), ("se", "8"), ("pe_order", "8")])
def test_archive_with_the_position_order_made_on_the_device(tmp_path, case, threads):
    """Round 4: the Pg-order export with the matched reads ordered on the device (ascending position, reads at one position
    by ascending index) instead of by the reference's sort, whose tie order only -t 1 reproduces.  HipReadsMatcher takes
    it when PgHelpers::numberOfThreads > 1 (or PGRC_DEVICE_SORT=1): the archive must decode to the input and have the size
    of the host-sort archive (the tie order moves a few bytes; at -t 8 the reference's parallel Pg generator and racy index
    move more)."""
    if not _have_e2e():
        pytest.skip("oracle/_ref was built without the encoder harness")
    env = {"PGRC_E2E_THREADS": threads}
    if threads == "1":
        env["PGRC_DEVICE_SORT"] = "1"
    r = _run(tmp_path, case, cpu_only=False, extra_env=env)
    assert r["gpu_gpu_calls"] >= 1 and r["roundtrip"], r
    if case in ("se", "pe"):
        assert r["gpu_device_exports"] >= 1, r
    tol = 0.0005 if threads == "1" else 0.01
    assert abs(r["gpu_bytes"] - r["cpu_bytes"]) <= tol * r["cpu_bytes"], r


@pytest.mark.gpu
def test_archive_identical_without_the_libstdcxx_shortcuts(tmp_path):
    """HipReadsMatcher's two libstdc++-dependent shortcuts (a string stream adopting the export arrays as its contents,
    pages of reserved vector storage touched ahead) are optimisations with guarded fallbacks: PGRC_NO_ADOPT /
    PGRC_NO_PRETOUCH take the fallbacks, the archive stays the same."""
    if not _have_e2e():
        pytest.skip("oracle/_ref was built without the encoder harness")
    r = _run(tmp_path, "se", cpu_only=False, extra_env={"PGRC_NO_ADOPT": "1", "PGRC_NO_PRETOUCH": "1"})
    assert r["gpu_gpu_calls"] >= 1 and r["gpu_device_exports"] >= 1 and r["identical"] and r["roundtrip"], r
