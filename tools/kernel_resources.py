#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of every kernel of one .hip source, from the assembly hipcc emits with the product flags.
   python tools/kernel_resources.py gemm.hip [substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pmgt_amd import _build  # noqa: E402


def main():
    src = os.path.join(_build.CSRC, sys.argv[1])
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    extra = sys.argv[3:]
    asm = subprocess.run([_build.HIPCC] + _build.HIP_FLAGS + extra + ["--cuda-device-only", "-S", src, "-o", "-"], capture_output=True, text=True)
    if asm.returncode != 0:
        sys.exit(asm.stderr)
    cur = None
    rows = {}
    for ln in asm.stdout.splitlines():
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", ln)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        if cur:
            m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|private_segment_fixed_size|group_segment_fixed_size|accum_offset)\s+(\S+)", ln)
            if m:
                rows[cur][m.group(1)] = m.group(2)
            if ".end_amdhsa_kernel" in ln:
                cur = None
    for k, v in rows.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        if sub and sub not in name:
            continue
        print(f"{name[:110]:110s} vgpr {v.get('next_free_vgpr'):>4s} sgpr {v.get('next_free_sgpr'):>4s} scratch {v.get('private_segment_fixed_size'):>5s} lds {v.get('group_segment_fixed_size'):>6s}")


if __name__ == "__main__":
    main()
