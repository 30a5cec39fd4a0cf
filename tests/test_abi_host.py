"""CPU-side checks of the product library: the C ABI loads and exports every symbol
include/zkmi.h declares, refuses to run without a gfx950 device, and its host-only pieces
(transcript, partial_verify) agree with the oracle.  No GPU compute here."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest

import golden_lib as G
import oracle_lib as O
import zk_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "zkmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(zk_amd.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libzkmi.so does not export {n}"
    # and the Python binding table covers the same set
    assert sorted(s[0] for s in zk_amd.SYMBOLS) == names


def test_no_device_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(zk_amd.ZkError) as ei:
        zk_amd.Context(0)
    assert ei.value.status == 9  # ZK_E_NO_DEVICE: no CPU fallback


def test_host_transcript_matches_oracle_and_kats():
    t = zk_amd.Transcript()
    assert t.peek_digest().hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    t.append(b"abc")
    assert t.peek_digest().hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    rng = random.Random(3)
    t, o = zk_amd.Transcript(), O.OTranscript()
    for n in (0, 1, 135, 136, 137, 32, 32, 1000, 271):
        data = bytes(rng.randrange(256) for _ in range(n))
        t.append(data)
        o.append(data)
        assert t.peek_digest() == o.peek_digest()
        assert list(t.sample_challenge()) == list(o.sample())
    got = t.sample_n_challenges(5)
    assert [list(r) for r in got] == [list(o.sample()) for _ in range(5)]
    # long appends take the whole-block path (several rate blocks per call): still the oracle's digest
    t, o = zk_amd.Transcript(), O.OTranscript()
    for n in (136 * 7, 136 * 3 + 5, 5000, 131):
        data = bytes(rng.randrange(256) for _ in range(n))
        t.append(data)
        o.append(data)
        assert t.peek_digest() == o.peek_digest()
    # the two builds of the round function (baseline ISA and BMI1/BMI2) agree; a process runs only one of them
    assert zk_amd.lib().zk_transcript_selftest(64) == 0


def test_host_partial_verify_matches_oracle_on_goldens():
    for case in G.vectors()["sum2x2"]:
        claim = O.fq1(G.unhex(case["claim"]))
        coeffs = np.stack([O.ints_to_mont([G.unhex(c) for c in g]) for g in case["round_polys"]])
        tr = zk_amd.Transcript()
        tr.append(case["transcript_prefix"].encode())
        ok, final, ch = zk_amd.SumcheckVerifier.partial_verify(zk_amd.SumCheckProof(claim, coeffs), tr)
        assert ok
        assert G.le32_of(final) == [case["final_claim"]]
        assert G.le32_of(ch) == case["challenges"]
        # a wrong claim fails in round 0 and reports the claim back (verifier.rs:41-45)
        bad = O.fq1(G.unhex(case["claim"]) + 1)
        ok, final, ch = zk_amd.SumcheckVerifier.partial_verify(zk_amd.SumCheckProof(bad, coeffs), zk_amd.Transcript())
        assert not ok and list(final) == list(bad)
    for case in G.vectors()["sumcheck"]:
        if not case["round_polys"]:
            continue
        claim = O.fq1(G.unhex(case["initial_claim_sum"]))
        coeffs = np.stack([O.ints_to_mont([G.unhex(c) for c in g]) for g in case["round_polys"]])
        table = G.case_table(case)
        tr = zk_amd.Transcript()
        tr.append(O.mle_to_bytes(table))
        ok, final, ch = zk_amd.SumcheckVerifier.partial_verify(zk_amd.SumCheckProof(claim, coeffs), tr)
        assert ok and G.le32_of(ch) == case["challenges"]
        otr = O.OTranscript()
        otr.append(O.mle_to_bytes(table))
        ok2, final2, _ = O.sumcheck_partial_verify(claim, coeffs, otr)
        assert ok2 and list(final) == list(final2)


def test_host_mirror_precondition_errors():
    class Fake:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

    with pytest.raises(zk_amd.ZkError) as ei:      # product_polynomial.rs:15-19
        zk_amd.ProductPolynomial([Fake(4), Fake(2)])
    assert ei.value.status == 3
    p4, p2 = zk_amd.ProductPolynomial([Fake(4), Fake(4)]), zk_amd.ProductPolynomial([Fake(2), Fake(2)])
    with pytest.raises(zk_amd.ZkError):            # sum_polynomial.rs:15-19
        zk_amd.SumPolynomial([p4, p2])
    s = zk_amd.SumPolynomial([p4, p4])
    assert s.degree() == 2 and s.number_of_variables() == 2 and s.length() == 4


def test_rust_sys_crate_is_in_step_with_the_header():
    """rust/zkmi-sys/src/lib.rs is generated from include/zkmi.h (tools/gen_rust_sys.py); it cannot be compiled here (no
    Rust toolchain), but it must declare exactly the functions the header declares, and be the generator's current output."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_rust_sys", os.path.join(ROOT, "tools", "gen_rust_sys.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text = open(os.path.join(ROOT, "rust", "zkmi-sys", "src", "lib.rs")).read()
    assert sorted(re.findall(r"pub fn (zk_[a-z0-9_]+)\(", text)) == _declared_functions()
    assert text == gen.generate(open(os.path.join(ROOT, "include", "zkmi.h")).read()), "run python3 tools/gen_rust_sys.py"


def test_rust_build_script_compiles_what_the_makefile_compiles():
    """rust/zkmi-sys/build.rs is the build a Rust maintainer runs; it cannot be run here (no cargo).  Its source list and
    compiler/linker flags must be those of zk_amd/csrc/Makefile, which IS exercised by build(): a translation unit added
    to one and not the other fails here (and, with -z defs, at link time on both sides)."""
    mk = open(os.path.join(ROOT, "zk_amd", "csrc", "Makefile")).read()
    rs = open(os.path.join(ROOT, "rust", "zkmi-sys", "build.rs")).read()
    mk_vars = {m.group(1): m.group(2).split() for m in re.finditer(r"^(\w+)\s*[:?]?=\s*(.*)$", mk, re.M)}
    rs_files = re.findall(r'"([^"]+)"', re.search(r"let files = \[(.*?)\];", rs, re.S).group(1))
    assert rs_files == mk_vars["SRCS"]
    for f in rs_files:
        assert os.path.exists(os.path.join(ROOT, "zk_amd", "csrc", f)), f
    rs_args = [a for grp in re.findall(r"\.args\(\[(.*?)\]\)", rs, re.S) for a in re.findall(r'"([^"]+)"', grp)]
    drop = {"-Wall", "-Wno-unused-result"}                       # diagnostics only
    mk_flags = [f for f in mk_vars["CXXFLAGS"] + mk_vars["LDFLAGS"] if f not in drop]
    assert sorted(a for a in rs_args if not a.startswith("--offload-arch") and a != "-shared") == sorted(mk_flags)
    assert "--offload-arch=gfx950" in rs_args and "-Wl,-z,defs" in rs_args
    assert mk_vars["ARCH"] == ["gfx950"]
