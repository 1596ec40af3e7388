"""Static check of the gfx950 ISA of every kernel file: no v_mfma may sit inside an EXEC-masked region that is not skipped
by a branch when the mask is empty.

Why: MFMA ignores EXEC.  When a block containing an MFMA is guarded by a condition the compiler considers DIVERGENT (anything
derived from threadIdx.x, e.g. a wave index `tid >> 6`, even though it is wave-uniform), the guard becomes
`s_and_saveexec` without a branch: the MFMA then runs in waves where the condition is false, while the VALU moves that
build its operands obey the mask.  That produced 1e27 bias gradients in the 256 x 256 weight-gradient tile (fixed by
`__builtin_amdgcn_readfirstlane` on the wave index).  Usage: python tools/audit_exec_masked_mfma.py  -> exit code 1 on a hit."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pmgt_amd import _build  # noqa: E402

FILES = [f for f in _build.HIP_SOURCES if f != "engine.hip"]
RESTORE = re.compile(r"s_or_b64 exec|s_mov_b64 exec|s_xor_b64 exec|s_or_saveexec|s_andn2_b64 exec|s_endpgm")


def isa(src, out):
    cmd = [_build.HIPCC] + _build.HIP_FLAGS + ["-S", "--cuda-device-only", os.path.join(_build.CSRC, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return out


def audit(path):
    hits = []
    func = None
    lines = open(path).read().split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func = m.group(1)
        if "s_and_saveexec" in l:
            skip = any("s_cbranch_execz" in x for x in lines[i + 1:i + 3])
            n = 0
            for lj in lines[i + 1:i + 400]:
                if RESTORE.search(lj):
                    break
                # MFMA, s_barrier and s_setprio all ignore EXEC; a barrier under a mask is a hang when the skip is missing
                n += any(op in lj for op in ("v_mfma", "s_barrier", "s_setprio"))
            if n and not skip:
                hits.append((func, i + 1, n))
    return hits


def main():
    bad = []
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(max_workers=6) as ex:
        outs = list(ex.map(lambda f: isa(f, os.path.join(tmp, f.replace(".hip", ".s"))), FILES))
        for f, o in zip(FILES, outs):
            for func, line, n in audit(o):
                bad.append(f"{f}: {func} (asm line {line}): {n} EXEC-ignoring instruction(s) (v_mfma / s_barrier / s_setprio) under an EXEC mask without a skip branch")
    for b in bad:
        print(b)
    print(f"audited {len(FILES)} files: {len(bad)} hit(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
