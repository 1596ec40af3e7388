"""Static ISA check (hipcc cross-compiles here): no MFMA under an EXEC mask without a skip branch -- see
tools/audit_exec_masked_mfma.py for the bug class this guards against."""
import os
import subprocess
import sys


def test_no_mfma_under_an_exec_mask():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "audit_exec_masked_mfma.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_every_lds_dma_function_waits_for_its_tiles_explicitly():
    """A tile brought by LDS-DMA (global_load_lds) is complete when the issuing wave's vmcnt says so -- a workgroup barrier alone does not
    wait for it (the compiler's __syncthreads() carries only the waits the memory model asks for: gemm_ws_kernel<16, MODE> read tiles in
    flight on the back edge of its unrolled loop, profiles/r04/NOTES.md section 8).  Source-level rule: every top-level function that issues
    an LDS-DMA also contains an explicit vmcnt wait (`s_waitcnt vmcnt(n)` or the counted `wait_vm(n)` helper)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "pmgt_amd", "csrc")
    checked = 0
    for name in sorted(os.listdir(csrc)):
        if not name.endswith(".hip"):
            continue
        lines = open(os.path.join(csrc, name)).read().split("\n")
        # top-level function bodies: from a column-0 line that opens one ("... {" without leading blank) to the next column-0 "}"
        starts = [i for i, l in enumerate(lines) if l and not l[0].isspace() and l.rstrip().endswith("{") and "(" in l and not l.startswith(("namespace", "struct", "enum", "extern", "//"))]
        for s in starts:
            e = next((j for j in range(s + 1, len(lines)) if lines[j].startswith("}")), len(lines) - 1)
            body = "\n".join(lines[s:e + 1])
            if "__builtin_amdgcn_global_load_lds" not in body:
                continue
            checked += 1
            assert re.search(r"s_waitcnt vmcnt\(|wait_vm\(", body), f"{name}:{s + 1}: LDS-DMA without an explicit vmcnt wait in the same function"
    assert checked >= 10, checked      # (the role-split kernels, the big tiles, the K = 512 streaming body, the full-row LayerNorm tile ...)
