"""The integer models the device field code is written against (no GPU): tools/field_model.py (the per-lane lazy-limb fields: every
intermediate inside its register, results against Python's integers) and tools/rows_field_model.py (the row-parallel k256 field of
csrc/ecgpu_rows.h with the DPP row-shift semantics measured on gfx950).  The device runs the same statements; its results are
compared with the oracle by tests/test_gpu_selftest.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tool):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_per_lane_field_model():
    assert "field model selftest: True" in run("field_model.py")


def test_row_parallel_k256_field_model():
    out = run("rows_field_model.py")
    assert "mul_rows: 2800 products equal" in out and "dbl_rows: 1200 doublings" in out


def test_p384_layout_model_runs():
    """tools/p384_layouts.py: the slot model behind the p384 limb layout (15 x 27) and the round-5 costing of one-level Karatsuba and
    signed 5-bit windows (neither reaches the 5 % that would justify building it)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "p384_layouts.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "15 x 27" in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "p384_layouts.py"), "--karatsuba-windows"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "Karatsuba" in r.stdout and "5-bit windows" in r.stdout
