"""Generates tests/golden/*.json from the pure-Python big-int oracle (oracle/pyref.py).

The reference (Rust, un-vendored ark-ff / sha3) cannot be built or run in this container
and holds no golden proofs of its own, so these vectors are outputs of
the independent Python restatement; the C oracle and the HIP path are both tested against
them.  Run:  python tests/golden/make_golden.py   (about a minute; output is committed)."""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyref  # noqa: E402

SEED = 0x5EED0000


def hx(v):
    return pyref.le32(v).hex()


def sumcheck_case(name, table, meta):
    pr = pyref.generate_sumcheck_proof(table)
    assert pyref.verify_proof(table, pr)
    return {"name": name, **meta, "initial_claim_sum": hx(pr["initial_claim_sum"]),
            "round_polys": [[hx(c) for c in g] for g in pr["round_polys"]],
            "challenges": [hx(c) for c in pr["challenges"]]}


def main():
    out = {"field": "BN254 Fq", "encoding": "LE32 hex of canonical values", "sumcheck": [], "sum2x2": [], "gkr": []}
    out["sumcheck"].append(sumcheck_case("ref_test_full_sumcheck_pass", [0, 0, 0, 3, 0, 0, 2, 5],
                                         {"table": [0, 0, 0, 3, 0, 0, 2, 5]}))
    for log_n in (1, 4, 10, 13, 16):
        table = pyref.gen_table(SEED, 0, 1 << log_n)
        out["sumcheck"].append(sumcheck_case(f"seeded_2^{log_n}", table, {"seed": SEED, "log_n": log_n}))

    # degree-2 sum of two products, 4 seeded tables (table_id k -> seed SEED+k), claim = true sum
    for log_n in (1, 3, 8):
        tabs = [pyref.gen_table(SEED + k, 0, 1 << log_n) for k in range(4)]
        f = [[tabs[0], tabs[1]], [tabs[2], tabs[3]]]
        claim = sum(pyref.sum_reduce(f)) % pyref.P
        tr = pyref.Transcript()
        tr.append(b"zk-amd golden sum2x2")
        proof, pts = pyref.generate_proof_for_partial_verify(claim, f, tr)
        tv = pyref.Transcript()
        tv.append(b"zk-amd golden sum2x2")
        ok, final, ch = pyref.partial_verify(proof, tv)
        assert ok and ch == pts
        assert final == sum(pyref.sum_reduce(pyref.sum_partial_evaluate(f, pts))) % pyref.P
        out["sum2x2"].append({"name": f"seeded_4x2^{log_n}", "seeds": [SEED + k for k in range(4)], "log_n": log_n,
                              "transcript_prefix": "zk-amd golden sum2x2", "claim": hx(claim),
                              "round_polys": [[hx(c) for c in g] for g in proof["round_polys"]],
                              "challenges": [hx(c) for c in pts], "final_claim": hx(final)})

    # GKR: the reference's test circuit, plus seeded layered circuits of uniform width 2^k
    ref_circuit = [[(0, 1, 0), (2, 3, 0), (4, 5, 0), (6, 7, 1)], [(0, 1, 1), (2, 3, 0)], [(0, 1, 0)]]
    cases = [("ref_test_gkr_sum_check", ref_circuit, list(range(1, 9)))]
    for k, depth in ((2, 3), (3, 2)):
        w = 1 << k
        rng = random.Random(1000 + k)
        layers = [[((2 * g) % w, (2 * g + 1) % w, rng.randrange(2)) for g in range(w)] for _ in range(depth)]
        cases.append((f"uniform_k{k}_d{depth}", layers, pyref.gen_table(SEED + 77, 0, w)))
    for name, layers, inputs in cases:
        proof = pyref.gkr_generate_proof(layers, pyref.Transcript(), inputs)
        assert pyref.gkr_verify_proof(inputs, layers, pyref.Transcript(), proof)
        out["gkr"].append({"name": name, "layers": layers, "inputs": [hx(v) for v in inputs],
                           "proof_bytes": pyref.gkr_proof_bytes(proof).hex(),
                           "rounds": [len(sp["round_polys"]) for sp in proof["sumcheck_proofs"]]})

    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(HERE, "vectors.json"))


if __name__ == "__main__":
    main()
