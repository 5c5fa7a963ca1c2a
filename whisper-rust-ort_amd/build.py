"""Builds libwhisper_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libwhisper_hip.so")
SOURCES = ["wh_mel.hip", "wh_gemm.hip", "wh_gemm8.hip", "wh_gemm8x.hip", "wh_gemm8_mx.hip", "wh_mlp.hip", "wh_attn.hip", "wh_decode.hip", "wh_dec_tile.hip", "wh_cross_es.hip", "wh_cross_es8.hip", "wh_cross_es3.hip", "wh_fp8.hip", "wh_model.cpp", "wh_api.cpp"]
HEADERS = ["wh_common.h", "wh_es_fp8.h", "wh_kernels.h", "wh_internal.h", "wh_json.h", "../../include/whisper_hip.h"]
# -amdgpu-mfma-vgpr-form: MFMA accumulators live in VGPRs (gfx950's register file is unified).  With the default
# heuristic the attention kernel kept its score and output tiles in AGPRs and spent 160 of ~400 VALU instructions
# per key tile on v_accvgpr_read/write copies around the softmax.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-pthread", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-x", "hip"]


# wh_attn.hip: the softmax maxima never see a NaN that matters (a NaN score poisons its row either way); without
# this every fmaxf operand is canonicalised first (v_max_f32 x, x, x), which doubles the max instructions of a loop
# that is VALU-bound.  Infinities keep their meaning (-inf masks the tail keys).
EXTRA_FLAGS = {"wh_attn.hip": ["-fno-honor-nans"], "wh_cross_es.hip": ["-fno-honor-nans"], "wh_cross_es8.hip": ["-fno-honor-nans"], "wh_cross_es3.hip": ["-fno-honor-nans"]}   # (the same for the running maxima of wh_cross_es.hip)


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose: bool = False, force: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]

    def compile_one(src: str) -> str:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src + ".o")
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_host(verbose, force)
    build_tools(verbose, force)
    return OUT


TOOLS = os.path.join(HERE, "..", "tools")


def build_tools(verbose: bool = False, force: bool = False) -> None:
    """Device-side check binaries the GPU tests run (tests/test_fp8_gpu.py, tests/test_hip_parity.py): the MX MFMA layout probe, the MX GEMM /
    LayerNorm kernels and the one-launch feed-forward block (k_enc_mlp) against host restatements.  They include the library's kernel source directly."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for name, kernel_src in (("mx_mfma_check", "wh_gemm8_mx.hip"), ("mx_gemm_check", "wh_gemm8_mx.hip"), ("mlp_check", "wh_mlp.hip"), ("es8_check", "wh_cross_es8.hip"), ("es3_check", "wh_cross_es3.hip")):
        src, exe = os.path.join(TOOLS, name + ".hip"), os.path.join(TOOLS, name)
        deps = [src, os.path.join(CSRC, kernel_src), os.path.join(CSRC, "wh_common.h"), os.path.join(CSRC, "wh_kernels.h")]
        if force or _stale(exe, deps):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-w", "-I", CSRC, src, "-o", exe]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)


HOST = os.path.join(HERE, "host")
HOST_SO = os.path.join(HERE, "libwh_host.so")
CLI = os.path.join(HERE, "whisper_bench")


def build_host(verbose: bool = False, force: bool = False) -> None:
    """The host side above the C ABI (C++, because the reference's host is compiled Rust and Rust is
    not in this image): the CLI `whisper_bench` and `libwh_host.so` (host helpers for the tests)."""
    gxx = os.environ.get("CXX", "g++")
    hdr = [os.path.join(HOST, "wh_host.h"), os.path.join(HOST, "wh_unicode_lower.h"), os.path.join(CSRC, "wh_json.h"), os.path.join(HERE, "..", "include", "whisper_hip.h")]
    src = os.path.join(HOST, "wh_host_capi.cpp")
    if force or _stale(HOST_SO, [src] + hdr):
        cmd = [gxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOST_SO, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    src = os.path.join(HOST, "whisper_bench.cpp")
    if force or _stale(CLI, [src, OUT] + hdr):
        cmd = [gxx, "-O2", "-std=c++17", "-Wall", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", CLI, src, "-L" + HERE, "-lwhisper_hip",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
