"""The wire-record codecs of the odd-sized parameter sets on the gfx950 ISA (tools/wire_codec_isa_check.py): whole words
(p521: one halfword + sixteen words) in, whole words out — no byte-wise loads for the compiler to merge and re-extract,
the code shape that once decoded p521 operands wrongly on the GPU (docs/DESIGN_long_form_r01-r05.md §4).  Compiles to assembly; no GPU needed.
The variable-base group of p224 and p521 (scalar and point records in, raw projective out) is checked here (half a minute);
the tool covers all four kernel groups and p192."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_p224_p521_wire_codecs_move_whole_words():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wire_codec_isa_check.py"), "--curve", "P224Params", "--curve",
                        "P521Params", "--groups", "var"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
    assert "k_var_base<ecgpu::P521Params, false>" in r.stdout and "halfword loads" in r.stdout          # the check saw the codec
