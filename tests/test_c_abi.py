"""The C ABI from a plain C program (gcc, no Python bindings, no C++): builds tests/c/abi_smoke.c against
include/recgraph_hip.h + librecgraph_hip.so.  CPU: it links, runs and reports RG_ERR_NO_DEVICE (exit 3).  GPU: its stdout is
the reference's stdout for the example data."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    from recgraph_amd import _lib
    _lib.build_library()
    out = str(tmp_path_factory.mktemp("cabi") / "abi_smoke")
    libdir = os.path.join(ROOT, "recgraph_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(HERE, "c", "abi_smoke.c"),
                           "-L" + libdir, "-lrecgraph_hip", "-Wl,-rpath," + libdir, "-o", out])
    return out


def _run(exe, mode):
    return subprocess.run([exe, os.path.join(HERE, "golden", "example_graph.gfa"), os.path.join(HERE, "golden", "example_reads.fa"), str(mode)],
                          capture_output=True, text=True, timeout=600)


def test_c_program_links_and_fails_loudly_without_a_device(exe):
    from recgraph_amd import _lib
    if _lib.load().rg_device_count() > 0:
        pytest.skip("a GPU is present")
    r = _run(exe, 8)
    assert r.returncode == 3 and r.stdout == "" and "align: -3" in r.stderr      # RG_ERR_NO_DEVICE, no CPU fallback


@pytest.mark.gpu
def test_c_program_prints_the_reference_stdout(exe, oracle, example_gfa, example_reads):
    names, reads = example_reads
    og = oracle.Graph.from_gfa_text(example_gfa)
    for mode, om in ((0, oracle.M0_SIMD), (2, oracle.M2), (4, oracle.M4_ABS), (8, oracle.M8_ABS)):
        r = _run(exe, mode)
        assert r.returncode == 0, r.stderr
        exp = "".join(og.align(om, rd, name=names[i], idx=i + 1)[0] for i, rd in enumerate(reads))
        assert r.stdout == exp, mode
        # rg_result_fields of read 0: modes 4 / 8 align it end to end; -m 0 / -m 2 may answer with the empty record of a
        # band failure (GAFStruct::new(): query_length 0)
        assert "has_record 1" in r.stderr and ("query_length 150" in r.stderr or mode in (0, 2))
