"""Build librsem_hip.so (HIP kernels + C ABI) and the CLI programs with hipcc, in-tree, for gfx950."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librsem_hip.so")
BIN = os.path.join(HERE, "bin")

HIP_SOURCES = ["status.hip", "comm.hip", "em.hip", "gibbs.hip", "model.hip", "ci.hip"]
HOST_PROGRAMS = {"rsem-run-em": ["host/run_em.cpp"], "rsem-run-gibbs": ["host/run_gibbs.cpp"],
                 "rsem-calculate-credibility-intervals": ["host/calc_ci.cpp"]}
# stages around the hot path that never touch the GPU: plain g++, no librsem_hip dependency
HOST_ONLY_PROGRAMS = {"rsem-parse-alignments": ["host/parse_alignments.cpp"]}
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               "-Wno-unused-result", "-Wno-unused-value", "-ffp-contract=off"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: librsem_hip.so cannot be built (there is no CPU fallback)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _headers():
    hs = [os.path.join(HERE, "..", "include", "rsem_hip.h")]
    for dp, _, fns in os.walk(CSRC):
        hs += [os.path.join(dp, f) for f in fns if f.endswith((".hpp", ".h"))]
    return hs


def build(force=False, verbose=False):
    cc = hipcc()
    objs = []
    hdrs = _headers()
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(HERE, "build", src.replace("/", "_") + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [cc] + HIPCC_FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    if force or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    os.makedirs(BIN, exist_ok=True)
    for prog, srcs in HOST_PROGRAMS.items():
        ss = [os.path.join(CSRC, s) for s in srcs]
        if not all(os.path.exists(s) for s in ss):
            continue
        out = os.path.join(BIN, prog)
        host_deps = ss + hdrs + [LIB]
        if force or _stale(out, host_deps):
            cmd = [cc, "-O2", "-std=c++17", "-o", out] + ss + ["-L" + HERE, "-lrsem_hip", "-Wl,-rpath,$ORIGIN/..", "-lpthread", "-lz"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    for prog, srcs in HOST_ONLY_PROGRAMS.items():
        ss = [os.path.join(CSRC, s) for s in srcs]
        out = os.path.join(BIN, prog)
        if force or _stale(out, ss + hdrs):
            cmd = ["g++", "-O2", "-std=c++17", "-o", out] + ss + ["-lpthread", "-lz"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print("built", LIB)
