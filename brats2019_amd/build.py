"""Build libresunet_hip.so (gfx950) in-tree with hipcc.  `python -m brats2019_amd.build [--force] [--dbg BITS]`.

`--dbg BITS` builds ANOTHER library, lib/libresunet_hip_dbg<BITS>.so, with -DRU_SB2_DBG=<BITS>: the ablation / section-counter
switches of conv3_sb2 (tools/sb2_*.sh, ru_dbg_sb2_prof) are compile-time and exist only there; select it with RU_LIB_PATH.  The
product library has none of them.

One `hipcc -c` per translation unit (run in parallel), then one link.  The shared library lands in
brats2019_amd/lib/ (git-ignored, but it travels to the GPU box with the gpurun snapshot).  Objects are
rebuilt only when a source or header is newer.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(PKG, "build")
LIB = os.path.join(LIBDIR, "libresunet_hip.so")
SOURCES = ["conv3_f32.hip", "conv3_f32c.hip", "conv3_sb.hip", "conv3_sb2_c16.hip", "conv3_sb2_c16_p1.hip", "conv3_sb2_mixed.hip", "conv3_wz.hip", "conv3_wz32.hip", "conv3_mx.hip", "conv3_wz32mx.hip", "wgrad_f32.hip", "wgrad_sb.hip", "wgrad_tr.hip", "pointwise.hip", "pointwise_c16.hip", "inference.hip", "engine.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC, "ru_common.h"), os.path.join(CSRC, "conv3_epilogue.hpp"), os.path.join(CSRC, "conv3_sb_common.hpp"), os.path.join(CSRC, "pw_helpers.hpp"), os.path.join(CSRC, "fin_tail.hpp"), os.path.join(CSRC, "conv3_wz.hpp"), os.path.join(CSRC, "conv3_wz_pack.hpp"), os.path.join(CSRC, "conv3_wz32.hpp"), os.path.join(CSRC, "conv3_mx.hpp"), os.path.join(CSRC, "conv3_mx_pack.hpp"), os.path.join(CSRC, "conv3_wz32mx.hpp"), os.path.join(os.path.dirname(PKG), "include", "resunet_hip.h")]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile(src, force, devtools=None):
    if not (src.startswith("conv3_sb") or src.startswith("conv3_wz") or src.startswith("conv3_mx") or src.startswith("wgrad_tr")):
        devtools = None                        # only this unit has RU_SB2_DBG switches: the others are shared with the product build
    obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + (".dbg%d.o" % devtools if devtools is not None else ".o"))
    path = os.path.join(CSRC, src)
    if not force and not _stale(obj, [path] + HEADERS):
        return obj, ""
    cmd = [_hipcc()] + FLAGS + (["-DRU_SB2_DBG=%d" % devtools] if devtools is not None else []) + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj, r.stderr


def build(force=False, verbose=True, devtools=None):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    lib = os.path.join(LIBDIR, "libresunet_hip_dbg%d.so" % devtools) if devtools is not None else LIB
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, devtools), srcs))
    objs = [o for o, _ in results]
    for _, warn in results:
        if warn.strip() and verbose:
            sys.stderr.write(warn)
    if force or _stale(lib, objs):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built %s (%d KB)" % (lib, os.path.getsize(lib) // 1024))
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, devtools=int(sys.argv[sys.argv.index("--dbg") + 1]) if "--dbg" in sys.argv else None)
