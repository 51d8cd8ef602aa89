"""BP_PROFILE=1 python scripts/prof_cfg3_phases.py -- the phases of bp_r1cs_prove at BASELINE config 3 on stderr (development aid)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_configs as BC

print(BC.driver_configs(small=False, reps=3)["cfg3_e2e"])
