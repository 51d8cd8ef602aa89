#!/usr/bin/env python3
"""Compare the output of integration/rust/pin_fixtures.rs (run against the REAL reference crate by someone with cargo) with what this
repository assumes -- the committed fixtures under tests/golden/ and the restatements in oracle/pyref.py.

    python3 scripts/compare_pins.py pins.jsonl

One line per pinned item: OK, or MISMATCH with the file(s) that encode the wrong assumption (DESIGN.md section 2 lists what to change).
Exit status 1 if anything mismatches.  Needs nothing but Python (no GPU, no built library)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")


def gold(name):
    with open(os.path.join(GOLD, name + ".json")) as f:
        return json.load(f)


def main():
    import pyref as R
    curves = {"bls12_381": R.BLS12_381, "bn254": R.BN254}
    bad = 0

    def report(item, curve, ok, where):
        nonlocal bad
        print("%-16s %-10s %s" % (item, curve, "OK" if ok else "MISMATCH -> " + where))
        bad += 0 if ok else 1

    for line in open(sys.argv[1]):
        line = line.strip()
        if not line:
            continue
        d = json.loads(line)
        item, cname = d["item"], d["curve"]
        c = curves[cname]
        mb = c.modbytes
        if item == "fr_to_bytes":
            want_one = (1).to_bytes(mb, "big").hex()
            want_big = (0x0102030405060708).to_bytes(mb, "big").hex()
            report(item, cname, d["modbytes"] == mb and d["one"] == want_one and d["x0102030405060708"] == want_big,
                   "bp_capi_ipp.hip commit_scalar, oracle/orc_ipp_tmpl.h fr_to_be, pyref.Transcript.commit_scalar (item 1)")
        elif item == "g1_to_bytes":
            G = c.g
            report(item, cname, d["generator"] == c.g1_to_bytes(G).hex() and d["identity"] == c.g1_to_bytes(None).hex() and
                   d["two_g"] == c.g1_to_bytes(c.add(G, G)).hex(), "BP_FMT_AMCL in bp_capi.hip, point_le_to_amcl in bp_capi_ipp.hip, pyref g1_to_bytes (item 2)")
        elif item == "fr_from_bytes":
            le = lambda v: (v % c.r).to_bytes(32, "little").hex()
            report(item, cname, d["all_ff"] == le((1 << (8 * mb)) - 1) and d["be_one"] == le(1),
                   "fr_from_be_reduce in bp_capi_ipp.hip, oracle/orc_ipp_tmpl.h, pyref challenge_scalar (item 3); le_one reported: " + d.get("le_one", "?"))
        elif item == "from_msg_hash":
            want = {x["msg"]: x["point"] for x in gold("hash_to_g1")["curves"][cname]["from_msg_hash"]}
            report(item + " " + (d["msg"] or '""'), cname, want.get(d["msg"]) == d["point"],
                   "bp_hash.cuh (k_hash_search / k_clear_cofactor), oracle/orc_api_tmpl.h from_msg_hash, pyref.g1_from_msg_hash (item 4)")
        elif item == "get_generators":
            want = gold("hash_to_g1")["curves"][cname]["get_generators"][d["prefix"]]
            report(item + " " + d["prefix"], cname, want == d["points"][:len(want)], "as from_msg_hash; also the counter starts at 1 (src/utils/mod.rs:18)")
        elif item == "generator":
            report(item, cname, d["G"] == gold("curves")[cname]["G"], "curve constants in bp_field.cuh / bp_curve.cuh / pyref.py (item 5: which BN254 this is); amcl prints " + d.get("G_hex", "?")[:60])
        elif item == "transcript":
            t = R.Transcript(b"pin")
            t.commit_point(c, b"P", c.g)
            t.commit_scalar(c, b"s", 5)
            ch = t.challenge_scalar(c, b"c")
            after = t.challenge_bytes(b"after", 32)
            report(item, cname, d["challenge"] == ch.to_bytes(32, "little").hex() and d["after"] == after.hex(),
                   "transcript framing / byte formats (items 1-3, 7): bp_merlin.hpp, bp_capi_ipp.hip, pyref.Transcript")
        elif item == "ipp":
            if "skipped" in d:
                print("%-16s %-10s skipped (%s)" % (item, cname, d["skipped"]))
                continue
            want = next(x for x in gold("ipp")[cname] if x["name"] == d["name"])
            ok = all(d[k] == want[k] for k in ("L", "R", "L_amcl", "a_out", "b_out", "transcript_after"))
            report(item, cname, ok, "everything above composed: tests/golden/ipp.json case " + d["name"])
        else:
            print("%-16s %-10s unknown item" % (item, cname))
    print("%d mismatch(es)" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
