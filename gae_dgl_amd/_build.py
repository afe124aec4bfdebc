"""Ahead-of-time build of libgae_hip.so for gfx950 (MI355X).

hipcc cross-compiles without a GPU; the shared library is kept in-tree
(``gae_dgl_amd/lib/``) so it travels with the repository snapshot."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(LIBDIR, "libgae_hip.so")
SOURCES = ["api.hip", "csr_build.hip", "plan_build.hip", "spmm.hip", "spmm_ell.hip", "dense.hip", "xw.hip", "tall.hip", "spfeat.hip", "decoder_bce.hip", "optim.hip", "readout.hip", "bce_dense.hip"]
ARCH = "gfx950"
# per-file extra flags.  decoder_bce: let MFMA accumulators live in VGPRs (gfx950 has a unified file) so the
# VALU epilogue of every tile does not pay one v_accvgpr_read per logit.
EXTRA_FLAGS = {"decoder_bce.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(ROOT, "include", "gae_hip.h"),
               os.path.join(ROOT, "include", "gae_hip_experimental.h")]
    headers += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.abspath(__file__))
    objs, rebuilt = [], False
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [sp] + headers):
            cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
                   "-I", os.path.join(ROOT, "include"), "-I", CSRC] + EXTRA_FLAGS.get(src, []) + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
            rebuilt = True
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if rebuilt or not os.path.exists(LIB):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(verbose=True))
