"""bench.py's one-line JSON contract, checked on a reduced size (2^14 points, 2 steps) so that it takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lg-n", "14", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]                      # stdout carries the one JSON line and nothing else
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "scalar-muls/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["verified"] is True                                   # the timed result equals the oracle's
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["kernel_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["matches_gpu"] is True and "sample" in c
    assert abs(d["value"] - (1 << 14) * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    assert "sweep" not in d                                        # the extras belong to the headline size (or --extras)


def test_bench_extras_sweep_and_strong_keys():
    """The driver runs ONE command per N; the line must therefore carry what north_star asks beside the headline:
    N = 1 -> "sweep" (sizes up to 2^22 in the real run; shrunk here), N > 1 -> "strong_2p22" (BASELINE config 4; shrunk, and on this
    one-GPU box in rehearsal mode).  Every entry is verified against the oracle."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lg-n", "14", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--extras",
                        "--sweep-max-lg", "17", "--configs-small"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    sw = d["sweep"]
    assert sorted(sw) == ["2^16", "2^17"]
    for e in sw.values():
        assert e["verified"] is True and e["ms"] > 0 and e["scalar_muls_per_s"] > 0 and e["algorithmic_GBs"] > 0
    # the other BASELINE configs and the PCIe-inclusive headline ride in the same line (shrunk here with --configs-small)
    h = d["with_scalar_h2d"]
    for k in ("pageable", "pinned"):
        assert h[k]["verified"] is True and h[k]["ms_per_step"] > 0 and h[k]["value"] > 0
    assert h["value"] == h["pinned"]["value"] and h["value"] < d["value"] * 1.5
    cf = d["configs"]
    assert cf["cfg1"]["n"] == 64 and cf["cfg1"]["proof_bit_exact_vs_oracle"] is True and cf["cfg1"]["create_ms"] > 0 and cf["cfg1"]["verify_ms"] > 0
    c3 = cf["cfg3_e2e"]
    assert c3["bytes_equal_oracle"] is True and c3["accepted"] is True and c3["tampered_rejected"] is True and c3["gates"] == 256
    assert c3["prove_ms"] > 0 and c3["verify_ms"] > 0 and c3["with_precomputed_generator_tables"]["same_proof_bytes"] is True
    c5 = cf["cfg5"]
    assert c5["msm_verified"] is True and c5["proof_bit_exact_vs_oracle"] is True and c5["msm_ms"] > 0 and c5["ipp_create_ms"] > 0 and c5["ipp_verify_ms"] > 0
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--lg-n", "14", "--steps", "2", "--warmup", "1", "--extras",
                        "--strong-lg", "16", "--rehearse-one-device"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert d["scaling"] == "weak" and d["verified"] is True and "sweep" not in d
    st = d["strong_2p16"]
    assert st["scaling"] == "strong" and st["n_total"] == 1 << 16 and st["n_per_gpu"] == 1 << 15 and st["verified"] is True
    assert st["value"] > 0 and st["speedup_vs_1gpu_same_n"] is None and st["shard_mode"] == "index"
    s2 = d["strong_2p16_index_x_windows"]                        # the 2-D mode (round 4): one index group, two window groups at N = 2
    assert s2["shard_mode"] == "index_x_windows" and (s2["index_groups"], s2["window_groups"]) == (1, 2) and s2["n_per_gpu"] == 1 << 16
    assert s2["verified"] is True and s2["value"] > 0


def test_bench_self_launches_n_ranks_and_strong_scaling():
    """`python bench.py --gpus 2 --strong` with no RANK in the environment starts the two ranks itself.  On this
    one-GPU box both ranks use GPU 0 and exchange records over gloo (--rehearse-one-device); the code path -- self-launch, index
    range split of a fixed total, record blocks with geometry headers, gather, finish over 2 sets, oracle check -- is the one
    the driver's N = 2, 4, 8 runs take with RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--strong", "--lg-n", "15", "--steps", "2", "--warmup", "1",
                        "--rehearse-one-device"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["rehearsal"] is True and d["verified"] is True
    assert d["config"]["n_total"] == 1 << 15 and d["config"]["n_per_gpu"] == 1 << 14
    # weak scaling keeps the per-GPU size
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--lg-n", "14", "--steps", "2", "--warmup", "1",
                        "--rehearse-one-device"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["n_total"] == 1 << 15 and d["verified"] is True
    # a mismatch between --gpus and an existing process group is an error, not a silent 1-GPU line
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--lg-n", "12", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                                                                              MASTER_PORT="29611"))
    assert p.returncode == 2 and not p.stdout.strip()
