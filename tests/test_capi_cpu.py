"""CPU: the C-ABI library loads, exports every symbol include/bpmsm.h declares, reports the right public
constants, and refuses to compute without a GPU (no CPU fallback)."""
import os
import re

import pytest

import __graft_entry__ as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bp():
    G.build()
    return G.load_package()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "bpmsm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bp_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(bp):
    names = declared_functions()
    assert len(names) >= 25
    lib = bp.lib()
    for n in names:
        assert hasattr(lib, n), "libbpmsm.so does not export " + n
        assert n in bp.SYMBOLS, "python mirror does not bind " + n
    for n in bp.SYMBOLS:
        assert n in names, n + " is bound but not declared in include/bpmsm.h"


def test_curve_params_match_golden(bp, golden):
    for name, cid in bp.CURVE_IDS.items():
        g = golden("curves")[name]
        info = bp.curve_info(cid)
        assert info.fp_bytes == 4 * g["fp_limbs32"] and info.fr_bytes == 32 and info.modbytes == g["modbytes"]
        assert int.from_bytes(bytes(info.p_le), "little") == int(g["p"], 16)
        assert int.from_bytes(bytes(info.r_le), "little") == int(g["r"], 16)
        assert info.fr_bits == int(g["r"], 16).bit_length()
        assert bytes(info.gen_le)[: 2 * info.fp_bytes].hex() == g["G"]
    assert bp.lib().bp_curve_params(7, None) == bp.BP_ERR_ARG


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback(bp):
    assert bp.device_count() == 0
    with pytest.raises(bp.DeviceError):
        bp.Context(bp.BLS12_381, 0)


def test_package_does_not_touch_the_oracle():
    """The product path may not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "bulletproofs-amcl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cuh", ".hip", ".cpp", ".h", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "oracle/" not in text.replace("the oracle", "") and "_oracle" not in text, os.path.join(dirpath, f)
