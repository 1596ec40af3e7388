#!/usr/bin/env python3
"""Instruction mix of attn_bwd_wgrad_kernel<16> per role and per phase, from the ISA the compiler emits.

    python tools/isa_mix.py [--kernel SUBSTR] [--out profiles/r04/abw_instruction_mix.md]

Builds pmgt_amd/csrc/attention_mfma.hip to assembly with -DPMGT_ABW_MARK (the product flags otherwise): the source then
drops `; ABW_MARK <role> <phase> ...` comment lines into the instruction stream (asm volatile comments: no instruction of
their own, though a marker is a scheduling fence for other volatile asm).  Every instruction between two markers is
attributed to the phase of the marker in front of it and classified by opcode:

  mfma            v_mfma_*
  valu.trans      v_exp / v_log / v_rcp / v_rsq / v_sqrt (quarter rate: 8 issue cycles)
  valu.fp32       v_add / v_sub / v_mul / v_fma / v_fmac / v_mad / v_max / v_min / v_med3 on f32 (the softmax, its backward, scaling)
  valu.dot        v_dot2* (norms of C)
  valu.cvt_pack   v_cvt_pk_bf16_f32 and other conversions (bf16 packing of images / outputs)
  valu.unpack     v_lshlrev_b32 / v_and_b32 with the bf16 <-> f32 patterns, v_perm, v_bfe, v_lshl_or (bf16 unpack / repack moves)
  valu.dpp        any VALU with a DPP modifier, v_permlane*, ds_swizzle (cross-lane reductions)
  valu.select     v_cndmask / v_cmp* (dropout predication, masks, diagonal)
  valu.int        v_mul_lo / v_mul_hi / v_mad_u* / v_xor / v_or / v_add_u32 / v_lshr / v_lshl / v_and on integers (dropout hash + address arithmetic)
  valu.mov        v_mov / v_accvgpr* / v_readlane / v_readfirstlane
  lds.read / lds.write   ds_read* / ds_write* (ds_read_b64_tr_b16 counted under lds.read)
  vmem            global_* / buffer_* / scratch_* (LDS-DMA, copy-out, mask loads)
  salu            s_* other than waits / barriers / nops / branches
  wait            s_waitcnt / s_nop / s_barrier / s_sleep / s_setprio
  branch          s_cbranch* / s_branch

Static counts of ONE pass through the phase (the loop bodies are straight-line per template instance); the roles run one
pass per step of two (sequence, head) pairs.  Cycle stamps per interval: tools/prof/abw_prof.py.
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]

CLASSES = ["mfma", "valu.trans", "valu.fp32", "valu.dot", "valu.cvt_pack", "valu.unpack", "valu.dpp", "valu.select", "valu.int", "valu.mov",
           "valu.other", "lds.read", "lds.write", "vmem", "salu", "wait", "branch"]


def classify(op, line):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        if "row_" in line or "quad_perm" in line or "row_shr" in line or op.startswith("v_permlane") or "dpp" in op:
            return "valu.dpp"
        if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
            return "valu.trans"
        if op.startswith("v_dot2"):
            return "valu.dot"
        if op.startswith("v_cvt"):
            return "valu.cvt_pack"
        if op.startswith("v_cndmask") or op.startswith("v_cmp"):
            return "valu.select"
        if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu.mov"
        if op.startswith(("v_perm_b32", "v_bfe", "v_lshl_or", "v_and_or", "v_pack", "v_bfi", "v_alignbit")):
            return "valu.unpack"
        if re.match(r"v_(add|sub|subrev|mul|fma|fmac|mad|mac|max|min|med3|max3|min3|pk_add|pk_mul|pk_fma|ldexp|fract|floor|trunc|rndne)_(f32|f16|legacy_f32)", op) or op.startswith(("v_fma_mix", "v_pk_")):
            return "valu.fp32"
        if re.match(r"v_(lshlrev|lshrrev|and|or|xor|not)_b(32|64)", op):
            # 16-bit shifts / 0xffff0000 masks are the bf16 <-> f32 moves; the rest is integer arithmetic (hash, addresses)
            if re.search(r"\b16\b", line.split(op, 1)[1].split(",")[1] if "," in line else "") or "0xffff0000" in line:
                return "valu.unpack"
            return "valu.int"
        if re.match(r"v_(mul_lo|mul_hi|mad_u|mad_i|mul_u|mul_i|add_u|add_co|addc|sub_u|sub_co|subrev_u|add3|xad|lshl_add|add_lshl|xor3|or3|and_or|lshlrev_b|ashrrev|min_u|max_u|min_i|max_i|mad_u64)", op):
            return "valu.int"
        return "valu.other"
    if op.startswith("ds_"):
        return "lds.write" if "write" in op or "store" in op else "lds.read"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio", "s_endpgm")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="attn_bwd_wgrad_kernelILi16E")
    ap.add_argument("--out", default="")
    ap.add_argument("--asm", default="", help="use this assembly file instead of compiling")
    args = ap.parse_args()
    asm = args.asm or "/tmp/pmgt_abw_mark.s"
    if not args.asm:
        src = os.path.join(ROOT, "pmgt_amd", "csrc", "attention_mfma.hip")
        subprocess.run([HIPCC] + FLAGS + ["-DPMGT_ABW_MARK", "-S", "--cuda-device-only", "-o", asm, src], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = open(asm).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and args.kernel in l and l.rstrip().endswith(":") or (args.kernel in l and l.split(":")[0].startswith("_ZN") and ":" in l and not l.startswith("\t")))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    phase = "prologue"
    mix = collections.OrderedDict()
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if "ABW_MARK" in t:
            phase = t.split("ABW_MARK", 1)[1].strip()
            mix.setdefault(phase, collections.Counter())
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c = classify(op, t)
        if c is None:
            continue
        mix.setdefault(phase, collections.Counter())[c] += 1
        mix[phase]["ops:" + op] += 1
    out = []
    out.append(f"# Instruction mix of `{args.kernel}` per role and phase (static ISA counts, one pass per step)\n")
    out.append("Generated by `tools/isa_mix.py` from the `-DPMGT_ABW_MARK` build of `pmgt_amd/csrc/attention_mfma.hip` (same flags as the product build). "
               "`attn ... it=I br=B`: attention wave of query / key tile I, softmax branch B (1 = cosine, 2 = scaled dot product); a workgroup runs "
               "two waves of each (one per sequence of the step) next to eight GEMM-role waves.  VALU issue costs 4 cycles (transcendentals 8), a "
               "16x16x32 MFMA holds the SIMD's vector issue for 8 of its 16 cycles (MI355X_MICROARCH.md).\n")
    hdr = ["phase"] + [c for c in CLASSES] + ["VALU total", "issue cycles (4 x VALU + 4 x trans + 8 x MFMA)"]
    out.append("| " + " | ".join(hdr) + " |")
    out.append("|" + "---|" * len(hdr))
    agg = collections.OrderedDict()
    for ph, cnt in mix.items():
        valu = sum(v for k, v in cnt.items() if k.startswith("valu."))
        cyc = 4 * valu + 4 * cnt["valu.trans"] + 8 * cnt["mfma"]
        out.append("| " + " | ".join([ph] + [str(cnt.get(c, 0)) for c in CLASSES] + [str(valu), str(cyc)]) + " |")
        role = ph.split()[0] + (" " + " ".join(p for p in ph.split() if p.startswith(("it=", "br="))) if ph.startswith("attn") else "")
        a = agg.setdefault(role, collections.Counter())
        for k, v in cnt.items():
            a[k] += v
    out.append("\n## Per role (sum over its phases of one step)\n")
    out.append("| role | " + " | ".join(CLASSES) + " | VALU total | VALU per MFMA | issue cycles |")
    out.append("|" + "---|" * (len(CLASSES) + 4))
    for role, cnt in agg.items():
        valu = sum(v for k, v in cnt.items() if k.startswith("valu."))
        cyc = 4 * valu + 4 * cnt["valu.trans"] + 8 * cnt["mfma"]
        out.append(f"| {role} | " + " | ".join(str(cnt.get(c, 0)) for c in CLASSES) + f" | {valu} | {valu / max(cnt['mfma'], 1):.1f} | {cyc} |")
    out.append("\n## Most frequent opcodes per role\n")
    for role, cnt in agg.items():
        ops = sorted(((v, k[4:]) for k, v in cnt.items() if k.startswith("ops:")), reverse=True)[:22]
        out.append(f"* **{role}**: " + ", ".join(f"`{k}` x{v}" for v, k in ops))
    text = "\n".join(out) + "\n"
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
