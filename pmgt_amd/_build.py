"""In-tree build of the native libraries (gfx950 only).

  pmgt_amd/lib/libpmgt_hip.so      HIP kernels + engine + C ABI   (hipcc --offload-arch=gfx950)
  pmgt_amd/lib/libpmgt_sampler.so  host MCNSampling               (g++)

hipcc cross-compiles without a GPU, so this runs in the dev container; the built .so files travel
to the GPU box with the repo snapshot (they are git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
OBJ = os.path.join(HERE, "csrc", "_obj")
HIP_SOURCES = ["gemm.hip", "gemm_ws.hip", "gemm_wsr.hip", "gemm_rowln.hip", "fp8.hip", "rowops.hip", "attention.hip", "attention_mfma.hip", "qkvc_attn.hip", "segsum.hip", "loss.hip", "optim.hip", "engine.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA accumulators stay in architectural VGPRs (no v_accvgpr_read moves before every VALU
# use of a result: -7 % VALU instructions in the attention backward, which is VALU-issue-bound)
# -fno-slp-vectorize: the SLP vectoriser pairs adjacent fp32 adds / multiplies of the epilogues into v_pk_add_f32 / v_pk_mul_f32,
# which issue no faster than the two scalar instructions they replace on gfx950 and cost v_mov shuffles to build the register
# pairs (streaming-GEMM LayerNorm epilogue: 485 -> 287 packed + move instructions per tile); the epilogues are VALU-bound:
# LayerNorm backward 1.09 -> 0.97 ms per step, step 11.00 -> 10.84 ms
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + os.environ.get("PMGT_EXTRA_HIP_FLAGS", "").split()


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))


def hip_lib_path():
    return os.path.join(LIB, "libpmgt_hip.so")


def sampler_lib_path():
    return os.path.join(LIB, "libpmgt_sampler.so")


def build_hip(force=False):
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "pmgt_capi.h"))
    jobs, objs = [], []
    for s in HIP_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + headers):
            jobs.append([HIPCC] + HIP_FLAGS + ["-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(_run, jobs))
    out = hip_lib_path()
    if force or jobs or _newer(out, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_sampler(force=False):
    os.makedirs(LIB, exist_ok=True)
    src = os.path.join(CSRC, "sampler.cpp")
    out = sampler_lib_path()
    hdr = os.path.join(os.path.dirname(HERE), "include", "pmgt_capi.h")
    if force or _newer(out, [src, hdr]):
        _run(["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", out, src])
    return out


def build_all(force=False):
    return build_hip(force), build_sampler(force)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
