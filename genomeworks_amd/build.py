"""In-tree build of the native libraries (hipcc for gfx950, no GPU needed).

  genomeworks_amd/lib/libgwhip.so            hand-written HIP kernels + the thin C-ABI (include/gwhip.h)
  genomeworks_amd/lib/libgenomeworks_amd.so  host C++ (Batch / Aligner, allocator, C API include/gw_capi.h)

Called by __graft_entry__.build(); also usable as `python -m genomeworks_amd.build`.
"""
import hashlib
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "lib")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

KERNEL_SRCS = ["csrc/gwhip_poa.hip", "csrc/gwhip_poa_part0.hip", "csrc/gwhip_poa_part1.hip", "csrc/gwhip_poa_part2.hip",
               "csrc/gwhip_poa_part3.hip", "csrc/gwhip_poa_part4.hip", "csrc/gwhip_poa_part5.hip", "csrc/gwhip_poa_part6.hip", "csrc/gwhip_poa_part7.hip", "csrc/gwhip_poa_hooks.hip", "csrc/gwhip_myers.hip", "csrc/gwhip_ukkonen.hip"]
HOST_SRCS = ["host/capi.cpp", "host/cudapoa_batch.cpp", "host/cudapoa_utils.cpp", "host/cudaaligner.cpp", "host/aligner_global.cpp", "host/device_pool.cpp",
             "host/alignment_impl.cpp", "host/runtime.cpp", "host/logging.cpp", "host/overlap_alignment.cpp", "host/multi_device.cpp"]

# no fast-math, no FMA contraction: band placement is IEEE fp32
KERNEL_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off",
                "-fhip-fp32-correctly-rounded-divide-sqrt"]
# experiments (same-box A/B of compiler options, tools/ab_headline.sh): extra hipcc flags for the kernel translation units
KERNEL_FLAGS += [f for f in os.environ.get("GW_KERNEL_EXTRA_FLAGS", "").split() if f]
HOST_FLAGS = ["-O2", "-g", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter",
              "-D__HIP_PLATFORM_AMD__", "-pthread"]


def _digest(paths, extra):
    h = hashlib.sha256(repr(extra).encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    return h.hexdigest()


def kernel_source_digest():
    """sha256 over the device sources (csrc/*.hip, *.h) and include/gwhip.h by repo-relative name and content: what a measurement
    session stamps into its PMC profile and bench.py recomputes at run time, so counters taken on other kernels are never
    attached to a line (the GPU box has no .git to ask)."""
    h = hashlib.sha256()
    base = os.path.join(PKG, "csrc")
    paths = [os.path.join(base, f) for f in os.listdir(base) if f.endswith((".hip", ".h"))] + [os.path.join(ROOT, "include", "gwhip.h")]
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.relpath(p, ROOT).encode())
            h.update(f.read())
    return h.hexdigest()


def _deps(subdir, exts):
    out = []
    for base in (os.path.join(PKG, subdir), os.path.join(ROOT, "include")):
        for d, _, fs in os.walk(base):
            out += [os.path.join(d, f) for f in fs if f.endswith(exts)]
    return out


def _run(cmd):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def _stale(target, stamp_value):
    stamp = target + ".stamp"
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != stamp_value


def _mark(target, stamp_value):
    with open(target + ".stamp", "w") as f:
        f.write(stamp_value)


def _local_includes(src):
    """Headers of csrc/ that `src` includes, transitively (plus include/gwhip.h): an object is rebuilt when its own
    source or one of these changes -- the POA translation unit takes minutes, the aligner ones seconds."""
    import re
    seen, todo = set(), [src]
    while todo:
        f = todo.pop()
        with open(f) as fh:
            for name in re.findall(r'#include\s+"([^"]+)"', fh.read()):
                path = os.path.normpath(os.path.join(os.path.dirname(f), name))
                if os.path.exists(path) and path not in seen:
                    seen.add(path)
                    todo.append(path)
    return sorted(seen)


def build_kernels(force=False):
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libgwhip.so")
    srcs = [os.path.join(PKG, s) for s in KERNEL_SRCS if os.path.exists(os.path.join(PKG, s))]
    objs, procs, sigs = [], [], []
    for s in srcs:
        o = os.path.join(LIB, os.path.basename(s) + ".o")
        objs.append(o)
        sig = _digest([s] + _local_includes(s), KERNEL_FLAGS)
        sigs.append(sig)
        if force or _stale(o, sig):
            cmd = [HIPCC] + KERNEL_FLAGS + ["-I", os.path.join(ROOT, "include"), "-c", s, "-o", o]
            print("[build]", " ".join(cmd), flush=True)
            procs.append((subprocess.Popen(cmd), o, sig))
    for p, o, sig in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
        _mark(o, sig)
    link_sig = _digest([], sigs)
    if force or procs or _stale(target, link_sig):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs)
        _mark(target, link_sig)
    return target


def build_host(force=False):
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libgenomeworks_amd.so")
    srcs = [os.path.join(PKG, s) for s in HOST_SRCS if os.path.exists(os.path.join(PKG, s))]
    sig = _digest(_deps("host", (".cpp", ".h", ".hpp")), HOST_FLAGS)
    if force or _stale(target, sig):
        cmd = ["g++"] + HOST_FLAGS + ["-shared", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROCM, "include"),
                                     "-o", target] + srcs + [
            "-L", LIB, "-lgwhip", "-L", os.path.join(ROCM, "lib"), "-lamdhip64",
            "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        _run(cmd)
        _mark(target, sig)
    return target


def build_cli(force=False):
    """The command-line tools -> genomeworks_amd/bin/: `cudapoa` (reference: cudapoa/src/main.cpp) and
    `align_overlaps` (the alignment stage of cudamapper, cudamapper/src/main.cu:54-187)."""
    bindir = os.path.join(PKG, "bin")
    os.makedirs(bindir, exist_ok=True)
    sig = _digest(_deps("host", (".cpp", ".h", ".hpp")), HOST_FLAGS)
    target = None
    for tool, main_src in (("align_overlaps", "align_overlaps_main.cpp"), ("cudapoa", "cudapoa_main.cpp")):
        target = os.path.join(bindir, tool)
        src = os.path.join(PKG, "host", main_src)
        if force or _stale(target, sig):
            cmd = ["g++"] + [f for f in HOST_FLAGS if f != "-fPIC"] + ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROCM, "include"),
                              "-o", target, src, "-L", LIB, "-lgenomeworks_amd", "-lgwhip", "-L", os.path.join(ROCM, "lib"),
                              "-lamdhip64", "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
            _run(cmd)
            _mark(target, sig)
    return target


def build_bindings(force=False):
    """The Cython package `genomeworks` (pygenomeworks/, API of the reference's pygenomeworks): three extension modules
    built in-tree against include/ and libgenomeworks_amd.so. Returns the package's parent directory (for sys.path)."""
    pkg = os.path.join(ROOT, "pygenomeworks")
    srcs = []
    for d, _, fs in os.walk(os.path.join(pkg, "genomeworks")):
        srcs += [os.path.join(d, f) for f in fs if f.endswith((".pyx", ".pxd"))]
    srcs += [os.path.join(pkg, "setup.py")] + _deps("host", (".hpp",))
    sig = _digest(srcs, "cython")
    target = os.path.join(pkg, "genomeworks", "bindings")  # stamp only
    import glob
    built = all(glob.glob(os.path.join(pkg, "genomeworks", m, m + ".*.so")) for m in ("cuda", "cudapoa", "cudaaligner"))
    stamp = target + ".stamp"
    fresh = built and os.path.exists(stamp) and open(stamp).read().strip() == sig
    if force or not fresh:
        _run_in(pkg, [sys.executable, "setup.py", "-q", "build_ext", "--inplace", "--force"])
        with open(stamp, "w") as f:
            f.write(sig)
    return pkg


def _run_in(cwd, cmd):
    print("[build] (in %s)" % cwd, " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=cwd, stdout=subprocess.DEVNULL)


def build_all(force=False):
    k = build_kernels(force)
    h = build_host(force)
    build_cli(force)
    build_bindings(force)
    return k, h


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
