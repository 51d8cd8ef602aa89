"""CPU: the product's host-side code that needs no GPU -- the Merlin transcript, the amcl byte formats it commits,
FieldElement::inverse and IPP::verification_scalars -- against the golden vectors and the oracle."""
import pytest

import __graft_entry__ as G
import _oracle as O


def hx(s):
    return bytes.fromhex(s)


@pytest.fixture(scope="module")
def bp():
    G.build()
    return G.load_package()


def test_transcript_matches_merlin_vectors(bp, golden):
    for c in golden("merlin"):
        t = bp.Transcript(hx(c["label"]))
        got = []
        for op in c["ops"]:
            if op[0] == "append":
                t.append_message(hx(op[1]), hx(op[2]))
            else:
                got.append(t.challenge_bytes(hx(op[1]), op[2]).hex())
        assert got == c["challenges"], c["name"]
    assert golden("merlin")[0]["challenges"][0] == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


@pytest.mark.parametrize("name", ["bls12_381", "bn254"])
def test_transcript_protocol_matches_oracle(bp, golden, name):
    cid = bp.CURVE_IDS[name]
    pts = [hx(c["sum"]) for c in golden("g1")[name]["add"]][:8]      # includes the identity
    sc = O.random_scalars(cid, 9, 4)
    t1, t2 = bp.Transcript(b"innerproduct"), O.Transcript(b"innerproduct")
    t1.append_u64(b"n", 64)
    t2.append_message(b"n", (64).to_bytes(8, "little"))
    for p in pts:
        t1.commit_point(cid, b"L", p)
        t2.commit_point(cid, b"L", p)
        assert t1.challenge_scalar(cid, b"u") == t2.challenge_scalar(cid, b"u")
    for i in range(4):
        s = sc[32 * i:32 * i + 32]
        t1.commit_scalar(cid, b"t_x", s)
        mb = 48 if cid == 0 else 32
        t2.append_message(b"t_x", int.from_bytes(s, "little").to_bytes(mb, "big"))       # FieldElement::to_bytes
        assert t1.challenge_bytes(b"c", 40) == t2.challenge_bytes(b"c", 40)
    # bp_transcript_commit_points(n) = n commit_point calls with the same label (the V commitments of a statement), oracle beside it
    t3, t4, t5 = bp.Transcript(b"r1cs"), bp.Transcript(b"r1cs"), O.Transcript(b"r1cs")
    t3.commit_points(cid, b"V", b"".join(pts), len(pts))
    for p in pts:
        t4.commit_point(cid, b"V", p)
        t5.commit_point(cid, b"V", p)
    z3, z4, z5 = t3.challenge_scalar(cid, b"z"), t4.challenge_scalar(cid, b"z"), t5.challenge_scalar(cid, b"z")
    assert z3 == z4 == z5
    t3.commit_points(cid, b"V", b"", 0)                                                   # no points: nothing absorbed
    assert t3.challenge_scalar(cid, b"y") == t4.challenge_scalar(cid, b"y")


@pytest.mark.parametrize("name", ["bls12_381", "bn254"])
def test_fr_inverse(bp, golden, name):
    cid = bp.CURVE_IDS[name]
    for c in golden("field")[name]["fr"][:60]:
        assert bp.fr_inverse(cid, hx(c["a"])) == hx(c["inv_a"])


@pytest.mark.parametrize("name", ["bls12_381", "bn254"])
def test_verification_scalars_match_oracle_and_golden(bp, golden, name):
    cid = bp.CURVE_IDS[name]
    for c in golden("ipp")[name]:
        n = c["n"]
        L = b"".join(hx(x) for x in c["L"])
        R = b"".join(hx(x) for x in c["R"])
        got = bp.IPP.verification_scalars(cid, L, R, n, bp.Transcript(b"innerproduct"))
        rc, want = O.ipp_verification_scalars(cid, O.Transcript(b"innerproduct"), L, R, len(c["L"]), n)
        assert rc == 0 and got == want, c["name"]
        with pytest.raises(bp.VerificationError):                     # n != 1 << lg_n  (src/ipp.rs:274-276)
            bp.IPP.verification_scalars(cid, L, R, 2 * n, bp.Transcript(b"innerproduct"))


def test_hostpool_alternating_runs_under_tsan(tmp_path):
    """HostPool::run must not let a helper carry a failed claim of one run into the next (larger) run.
    tests/cpp/hostpool_stress.cpp alternates run(4) / run(8) with tiny jobs; built with ThreadSanitizer, every job runs exactly once."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "hostpool_stress")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", os.path.join(root, "tests", "cpp", "hostpool_stress.cpp"), "-o", exe])
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}      # (scripts/sanitize_cpu.sh preloads the ASan runtime: not into a TSan binary)
    p = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "hostpool_stress ok" in p.stdout, p.stdout + p.stderr
    assert "ThreadSanitizer" not in p.stderr, p.stderr
