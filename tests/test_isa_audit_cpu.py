"""Static ISA check (hipcc cross-compiles here): no MFMA under an EXEC mask without a skip branch -- see
tools/audit_exec_masked_mfma.py for the bug class this guards against."""
import os
import subprocess
import sys


def test_no_mfma_under_an_exec_mask():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "audit_exec_masked_mfma.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
