#!/usr/bin/env python3
"""Opcode histogram of a range of lines of a gfx950 assembly listing (hipcc -S --cuda-device-only).

usage: isa_hist.py file.s [first:last ...]      (1-based line numbers of the listing; default: the whole file)
Prints, per range, the instruction count per opcode (descending) and the totals per class:
  half = instructions gfx950 issues at half rate (microbench/instr_rate.hip): 64-bit multiplies / mads, v_mul_lo/hi, 64-bit shifts,
         every carry-in / carry-out addition, v_mad_u32_u24, v_add3_u32
  full = other VALU; salu; mem (global / ds / scratch / buffer); other
"""
import collections
import re
import sys

HALF = re.compile(r"^v_(mad_u64_u32|mad_i64_i32|mul_lo_u32|mul_hi_u32|mul_hi_i32|lshrrev_b64|lshlrev_b64|ashrrev_i64|add_co_u32|addc_co_u32|"
                  r"sub_co_u32|subb_co_u32|subrev_co_u32|subbrev_co_u32|lshl_add_u64|mad_u32_u24|add3_u32|mul_u32_u24|fma_f64)")


def classify(op):
    if op.startswith("v_"):
        return "half" if HALF.match(op) else "full"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "ds_", "scratch_", "buffer_", "flat_")):
        return "mem"
    return "other"


def hist(lines):
    h = collections.Counter()
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":") or re.match(r"^[.\w$]+:", s):
            continue
        h[s.split()[0]] += 1
    return h


def main():
    path = sys.argv[1]
    with open(path) as f:
        lines = f.readlines()
    ranges = sys.argv[2:] or ["1:%d" % len(lines)]
    for r in ranges:
        a, b = (int(x) for x in r.split(":"))
        h = hist(lines[a - 1:b])
        cls = collections.Counter()
        for op, n in h.items():
            cls[classify(op)] += n
        print("== lines %d..%d: %d instructions  %s" % (a, b, sum(h.values()), dict(cls)))
        for op, n in h.most_common():
            print("  %6d  %-28s %s" % (n, op, classify(op)))


if __name__ == "__main__":
    main()
