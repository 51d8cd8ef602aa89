import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G
import _oracle as O
bp = G.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "bn254"
cases = json.load(open(os.path.join(ROOT, "tests/golden/ipp.json")))[name]
ctx = bp.Context(bp.CURVE_IDS[name], 0)
cid = ctx.curve
hx = bytes.fromhex
for c in cases[1:3]:
    n = c["n"]
    print("case", c["name"], n, flush=True)
    cat = lambda k: b"".join(hx(x) for x in c[k])
    Gv = bp.G1Vector.from_bytes(ctx, cat("G"), n); Hv = bp.G1Vector.from_bytes(ctx, cat("H"), n)
    Gf = bp.FieldElementVector.from_bytes(ctx, cat("G_factors"), n); Hf = bp.FieldElementVector.from_bytes(ctx, cat("H_factors"), n)
    a = bp.FieldElementVector.from_bytes(ctx, cat("a"), n); b = bp.FieldElementVector.from_bytes(ctx, cat("b"), n)
    st = bp.IPPState(ctx, Gv, Hv, hx(c["Q"]), Gf, Hf, a, b)
    tr = O.Transcript(b"innerproduct")
    tr.append_message(b"dom-sep", b"ipp v1"); tr.append_message(b"n", n.to_bytes(8, "little"))
    k = 0
    while len(st) > 1:
        L, R = st.round()
        print(" round", k, "L ok", L == hx(c["L"][k]), "R ok", R == hx(c["R"][k]), flush=True)
        tr.commit_point(cid, b"L", L); tr.commit_point(cid, b"R", R)
        u = tr.challenge_scalar(cid, b"u")
        ui = bp.fr_inverse(cid, u)
        print(" u", u.hex(), "ui", ui.hex(), flush=True)
        st.fold(u, ui)
        print(" folded", flush=True)
        k += 1
    print(" finish", st.finish()[0] == hx(c["a_out"]), flush=True)
