#!/usr/bin/env python3
"""Per-kernel resource table of the gfx950 code objects inside libbpmsm.so.

usage: kernel_resources.py [path/to/libbpmsm.so] [--check]
Unbundles every hipv4-amdgcn-amd-amdhsa--gfx950 code object with clang-offload-bundler, reads the AMDGPU metadata note
(llvm-readelf --notes) and prints, per kernel: VGPRs, AGPRs, SGPRs, LDS bytes, scratch bytes, spilled VGPRs / SGPRs.
--check: exit 1 if any kernel spills a VGPR (tests/test_capi_cpu.py runs it that way; profiles/r04_kernel_resources.txt is its output).
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(so, tmp):
    """The .hip_fatbin section of a hipcc shared object holds one clang-offload-bundle per translation unit."""
    raw = os.path.join(tmp, "fatbin.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, raw])
    data = open(raw, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    outs = []
    pos = [m.start() for m in re.finditer(re.escape(magic), data)]
    for k, p in enumerate(pos):
        end = pos[k + 1] if k + 1 < len(pos) else len(data)
        bundle = os.path.join(tmp, "bundle%d.bin" % k)
        open(bundle, "wb").write(data[p:end])
        out = os.path.join(tmp, "co%d.elf" % k)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + bundle,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(out) and os.path.getsize(out) > 0:
            outs.append(out)
    return outs


def kernels_of(elf):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], capture_output=True, text=True, check=True).stdout
    rows = []
    cur = None
    for ln in txt.splitlines():
        s = ln.strip()
        if s.startswith("- .agpr_count:") or s.startswith("- .args:"):
            cur = {}
            rows.append(cur)
        m = re.match(r"-?\s*\.(\w+):\s+(.*)$", s)
        if m and cur is not None:
            k, v = m.group(1), m.group(2).strip()
            if k in ("agpr_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count",
                     "sgpr_spill_count", "name", "max_flat_workgroup_size"):
                cur[k] = v.strip("'")
    return [r for r in rows if "name" in r]


def demangle(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return p.stdout.splitlines() if p.returncode == 0 else names
    except OSError:
        return names


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    check = "--check" in sys.argv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = args[0] if args else os.path.join(root, "bulletproofs-amcl_amd", "libbpmsm.so")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(so, tmp):
            rows += kernels_of(co)
    names = demangle([r["name"] for r in rows])
    for r, nm in zip(rows, names):
        nm = re.sub(r"\(.*$", "", nm).replace("void ", "").replace("bp::", "")
        r["short"] = nm
    rows.sort(key=lambda r: r["short"])
    print("%-46s %5s %5s %5s %7s %8s %7s %7s" % ("kernel (gfx950)", "vgpr", "agpr", "sgpr", "lds B", "scratch", "vspill", "sspill"))
    bad = 0
    for r in rows:
        g = lambda k: int(r.get(k, "0") or 0)
        print("%-46s %5d %5d %5d %7d %8d %7d %7d" % (r["short"][:46], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                                                    g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
        bad += 1 if g("vgpr_spill_count") > 0 else 0
    print("%d kernels, %d with spilled VGPRs" % (len(rows), bad))
    if check and bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
