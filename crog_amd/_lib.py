"""Build and bind libcrog_hip.so (the gfx950 kernel library) through its C ABI.

The prototypes are read from include/crog_hip.h itself, so the header is the single source of
truth for the boundary: every declared symbol must exist in the .so (checked at load) and the
ctypes argtypes are derived from the C declarations.

There is deliberately no CPU fallback: if the library cannot be loaded, importing the compute
path raises (tests that need no GPU only use `parse_header` / `load`).
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(ROOT)
CSRC = os.path.join(ROOT, "csrc")
HEADER = os.path.join(REPO, "include", "crog_hip.h")
LIB_PATH = os.environ.get("CROG_LIB") or os.path.join(ROOT, "libcrog_hip.so")
BUILD_DIR = os.path.join(ROOT, "csrc", "build")
SOURCES = ["api.hip", "gemm.hip", "gemm_pp.hip", "gemm_ppt.hip", "conv_sw.hip", "wgrad_sw.hip", "gemm_skinny.hip", "norm.hip", "eltwise.hip", "head.hip", "conv_aux.hip", "attn.hip", "ssg.hip", "preprocess.hip", "replay.hip", "comm.hip"]
# NO_PACKED_F32: the device code is built WITHOUT the packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: what the
# SLP vectoriser makes of adjacent scalar fp32 operations - 19 880 of them in 236 kernels under plain -O3).  Round 5 found their results wrong
# in lanes 48-63 of a wave when an MFMA kernel of ANOTHER stream shares the SIMD: the bilinear x2 backward beside a weight-gradient GEMM
# differed from its serial result in 2997 of 3000 launches (scripts/pk_probe.py), 0 of 3000 without the instructions, and deterministic mode
# with every side stream went from 11 / 11 differing passes to 0 / 23 (LAB_NOTES section 10).  The host half of the compile does not know
# the feature and says so ("not a recognized feature for this target (ignoring feature)"): harmless.  scripts/count_pk.py counts them;
# tests/test_abi_host.py checks that a rebuilt source holds none.
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc"] + NO_PACKED_F32 + ["-DCROG_NO_PACKED_F32=1"]      # (the macro is what crog_build_flags() reports)
# norm.hip (BatchNorm / LayerNorm / softmax) is compiled WITHOUT fused-multiply-add contraction: its kernels are HBM-bound, and its fp32
# arithmetic then has the reference's structure - every product rounded, as torch's kernels form them.  y = z * scale + (beta - mean * scale)
# relies on fl(z * scale) and fl(mean * scale) rounding alike when z is close to the channel's mean (two samples per channel in config 1's
# BatchNorm1d); contracted, the exact z * scale meets the rounded mean * scale.  Measured on config 1 at B = 2 (tests/test_fulldepth_gpu.py):
# logits 2.6-2.8e-3 from the float64 result with contraction, 0.7-0.9e-3 without (the reference's own fp32: 1.3e-3).
EXTRA_FLAGS = {"norm.hip": ["-ffp-contract=off"]}


class GemmDesc(ctypes.Structure):
    """Mirror of `crog_gemm_desc` (include/crog_hip.h)."""

    _fields_ = [
        ("dtype", ctypes.c_int), ("a_layout", ctypes.c_int), ("b_layout", ctypes.c_int),
        ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p),
        ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int),
        ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64), ("ldc", ctypes.c_int64),
        ("batch", ctypes.c_int), ("batch_inner", ctypes.c_int),
        ("sAo", ctypes.c_int64), ("sAi", ctypes.c_int64), ("sBo", ctypes.c_int64),
        ("sBi", ctypes.c_int64), ("sCo", ctypes.c_int64), ("sCi", ctypes.c_int64),
        ("splitk", ctypes.c_int),
        ("convH", ctypes.c_int), ("convW", ctypes.c_int), ("convC", ctypes.c_int),
        ("alpha", ctypes.c_float), ("bias", ctypes.c_void_p), ("act", ctypes.c_int),
        ("R", ctypes.c_void_p), ("ldr", ctypes.c_int64), ("out_mode", ctypes.c_int), ("debug", ctypes.c_int),
        ("col_stats", ctypes.c_void_p), ("stat_replicas", ctypes.c_int), ("a_sum", ctypes.c_void_p),
        ("bwd_z", ctypes.c_void_p), ("ldz", ctypes.c_int64), ("bwd_ss", ctypes.c_void_p), ("bwd_mask", ctypes.c_void_p),
        ("stat_sync", ctypes.c_void_p),
    ]


_CTYPE = {
    "int": ctypes.c_int, "float": ctypes.c_float, "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64, "crog_stream_t": ctypes.c_void_p, "size_t": ctypes.c_size_t,
}


def parse_header(path: str = HEADER):
    """Return {name: (restype, [argtypes], [argnames])} for every function declared in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"typedef struct crog_gemm_desc \{.*?\} crog_gemm_desc;", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?:^|[;}\n])\s*(const char\s*\*|int)\s+(crog_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
        argtypes, argnames = [], []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                argnames.append(a.split()[-1].lstrip("*"))
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = a.replace("const ", "").split()[0]
                    argtypes.append(_CTYPE[ty])
        protos[name] = (restype, argtypes, argnames)
    return protos


def _needs_rebuild(obj: str, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose: bool = False, force: bool = False) -> str:
    """Compile every HIP source for gfx950 and link crog_amd/libcrog_hip.so (in-tree)."""
    os.makedirs(BUILD_DIR, exist_ok=True)
    common = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_dma.h"), os.path.join(CSRC, "comm_dev.h"), HEADER,
              os.path.abspath(__file__)]      # (this file holds the compile flags)
    hipcc = os.environ.get("HIPCC", "hipcc")

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(BUILD_DIR, src.replace(".hip", ".o"))
        if force or _needs_rebuild(o, [s] + common):
            cmd = [hipcc] + HIPCC_FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return o

    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _needs_rebuild(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB_PATH


_lib = None


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    """dlopen the library and bind every prototype of the header. Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (torch's bundled HIP runtime must be the one already resident when the library is dlopened)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(crog_amd has no CPU fallback)")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes, _) in parse_header().items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"libcrog_hip.so does not export {name} declared in crog_hip.h") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if not packed_f32_off(lib):
        # a library supplied through CROG_LIB, or built by other means: its fp32 element-wise kernels may hold v_pk_*_f32, whose
        # results were found wrong beside another stream's MFMA kernel (NO_PACKED_F32 above).  runtime.set_deterministic refuses such a library.
        import warnings
        warnings.warn(f"{path} was built with packed-fp32 VALU instructions ({lib.crog_build_flags().decode()}): results beside a second "
                      "stream's GEMM are not trustworthy on gfx950 and deterministic mode is unavailable; rebuild with crog_amd._lib.build()")
    _lib = lib
    return lib


def packed_f32_off(lib=None) -> bool:
    """Was the loaded library compiled without the packed-fp32 VALU instructions (crog_build_flags)?"""
    lib = lib if lib is not None else load()
    return b"packed-fp32-ops=off" in lib.crog_build_flags()


def source_digest() -> str:
    h = hashlib.sha1()
    for f in SOURCES + ["common.h", "gemm_dma.h", "comm_dev.h"]:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(HEADER, "rb").read())
    return h.hexdigest()[:12]


class CrogError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().crog_last_error().decode()
        raise CrogError(f"{what}: rc={rc}: {msg}")
