"""In-tree build of the native extensions.

Two shared objects are produced next to this file:

* ``_rt``  – C++17 runtime (graph scheduler, blocks, record files, snapshot,
  logging, channel, timer), no GPU dependency, compiled with g++.
* ``_hip`` – every HIP kernel under ``csrc/hip`` compiled by hipcc for gfx950
  plus a pybind11 binding file that passes raw device pointers and a stream.

Run ``python -m singa_amd.build`` (add ``--force`` to rebuild everything).
"""
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
ARCH = "gfx950"
EXT_SUFFIX = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
# extra hipcc flags for diagnostic builds, e.g. SA_HIPCC_FLAGS=-DSA_CONV_PROF
EXTRA_HIPCC = os.environ.get("SA_HIPCC_FLAGS", "").split()


def _py_includes():
    import pybind11
    return ["-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"]]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _headers(*dirs):
    out = []
    for d in dirs:
        for root, _, files in os.walk(d):
            out += [os.path.join(root, f) for f in files
                    if f.endswith((".h", ".hpp", ".cuh"))]
    return out


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _compile_many(jobs, workers):
    with ThreadPoolExecutor(max_workers=workers) as ex:
        for out in ex.map(_run, jobs):
            if out.strip():
                sys.stderr.write(out)


def build_rt(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    rt = os.path.join(CSRC, "runtime")
    srcs = [os.path.join(rt, f) for f in sorted(os.listdir(rt)) if f.endswith(".cc")]
    srcs.append(os.path.join(CSRC, "bindings", "rt_bind.cc"))
    hdrs = _headers(rt)
    target = os.path.join(HERE, "_rt" + EXT_SUFFIX)
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(OBJ, "rt_" + os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append(["g++", "-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                         "-Wall", "-c", s, "-o", o] + _py_includes())
    if jobs and verbose:
        print("[singa_amd.build] compiling %d runtime sources" % len(jobs))
    _compile_many(jobs, 8)
    if jobs or not os.path.exists(target):
        _run(["g++", "-shared", "-o", target] + objs + ["-lpthread"])
    return target


def build_hip(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hd = os.path.join(CSRC, "hip")
    srcs = [os.path.join(hd, f) for f in sorted(os.listdir(hd)) if f.endswith(".hip")]
    hdrs = _headers(hd)
    target = os.path.join(HERE, "_hip" + EXT_SUFFIX)
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(OBJ, "hip_" + os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17",
                         "-fPIC", "-ffast-math", "-fno-gpu-rdc",
                         "-Wno-unused-result"] + EXTRA_HIPCC +
                        ["-c", s, "-o", o])
    bind = os.path.join(CSRC, "bindings", "hip_bind.cc")
    bo = os.path.join(OBJ, "hip_bind.o")
    objs.append(bo)
    if force or _newer(bo, [bind] + hdrs):
        jobs.append(["g++", "-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                     "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include"),
                     "-I" + hd, "-c", bind, "-o", bo] + _py_includes())
    if jobs and verbose:
        print("[singa_amd.build] compiling %d HIP sources for %s" % (len(jobs), ARCH))
    _compile_many(jobs, 8)
    if jobs or not os.path.exists(target):
        _run(["g++", "-shared", "-o", target] + objs +
             ["-L" + os.path.join(ROCM, "lib"), "-lamdhip64",
              "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-lpthread", "-ldl"])
    return target


def _embed_opencl_sources(cl_dir):
    """Write the directory's .cl files as one C string literal sequence into
    ``cl_sources.inc`` (included by opencl/runtime.cc, which hands the text to
    the OpenCL compiler at run time).  Role of the reference's
    tool/opencl/clsrc_to_str.py.  Returns the .cl paths."""
    files = sorted(os.path.join(cl_dir, f) for f in os.listdir(cl_dir)
                   if f.endswith(".cl"))
    lines = []
    for path in files:
        lines.append('"// ---- %s\\n"' % os.path.basename(path))
        with open(path) as f:
            for line in f.read().splitlines():
                lines.append('"%s\\n"' % line.replace("\\", "\\\\")
                             .replace('"', '\\"'))
    text = "\n".join(lines) + "\n"
    out = os.path.join(cl_dir, "cl_sources.inc")
    old = open(out).read() if os.path.exists(out) else None
    if old != text:
        with open(out, "w") as f:
            f.write(text)
    return files


def build_cpp(force=False, verbose=True):
    """C++ API: ``lib/libsinga_amd_cpp.so`` (device / memory pool / tensor /
    layers / FeedForwardNet / optimizers / IO / network over the same HIP
    kernels) and its self-checking test program ``bin/singa_cpp_test``."""
    build_hip(force, verbose)  # the kernel objects are linked into the library
    os.makedirs(OBJ, exist_ok=True)
    cpp = os.path.join(CSRC, "cpp")
    inc = os.path.join(cpp, "include")
    rt = os.path.join(CSRC, "runtime")
    hd = os.path.join(CSRC, "hip")
    cl_sources = _embed_opencl_sources(os.path.join(cpp, "src", "core", "opencl"))
    hdrs = _headers(inc, rt, hd) + cl_sources + [
        os.path.join(cpp, "src", "core", "opencl", "runtime.h")]
    flags = ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-parameter",
             "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include"),
             "-I" + inc]
    srcs = []
    for root, _, files in os.walk(os.path.join(cpp, "src")):
        srcs += [os.path.join(root, f) for f in sorted(files) if f.endswith(".cc")]
    # the runtime pieces the C++ API re-exports (default symbol visibility)
    srcs += [os.path.join(rt, f) for f in ("graph.cc", "logging.cc", "snapshot.cc")]
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(OBJ, "cpp_" + os.path.relpath(s, CSRC).replace(os.sep, "_") + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append(flags + ["-c", s, "-o", o])
    if jobs and verbose:
        print("[singa_amd.build] compiling %d C++ API sources" % len(jobs))
    _compile_many(jobs, 8)
    hip_objs = [os.path.join(OBJ, "hip_" + f + ".o")
                for f in sorted(os.listdir(hd)) if f.endswith(".hip")]
    libdir = os.path.join(HERE, "lib")
    bindir = os.path.join(HERE, "bin")
    os.makedirs(libdir, exist_ok=True)
    os.makedirs(bindir, exist_ok=True)
    lib = os.path.join(libdir, "libsinga_amd_cpp.so")
    if jobs or not os.path.exists(lib) or _newer(lib, hip_objs):
        _run(["g++", "-shared", "-o", lib] + objs + hip_objs +
             ["-L" + os.path.join(ROCM, "lib"), "-lamdhip64",
              "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-lpthread"])
    outs = [lib]
    tests = [("singa_cpp_test", os.path.join(cpp, "test", "test_main.cc"))]
    ex = os.path.join(HERE, "..", "examples", "cpp")
    if os.path.isdir(ex):
        for root, _, files in os.walk(ex):
            for f in sorted(files):
                if f.endswith(".cc"):
                    tests.append(("singa_" + os.path.splitext(f)[0].replace("-", "_"),
                                  os.path.join(root, f)))
    for name, src in tests:
        exe = os.path.join(bindir, name)
        if force or _newer(exe, [src, lib] + hdrs):
            _run(flags + [src, "-o", exe, "-L" + libdir, "-lsinga_amd_cpp",
                          "-Wl,-rpath,$ORIGIN/../lib",
                          "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-lpthread"])
        outs.append(exe)
    return outs


SANITIZE_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                  "-fno-omit-frame-pointer", "-O1", "-g"]


def build_sanitized(outdir, verbose=True):
    """Host-side code under AddressSanitizer + UndefinedBehaviorSanitizer:
    the runtime extension (``<outdir>/_rt<suffix>``), the C++ API library
    (``<outdir>/lib``) and its test program (``<outdir>/bin/singa_cpp_test``).
    Only host translation units are instrumented; the HIP kernel objects are
    linked as built by ``build_hip`` (device code cannot be sanitized here).
    The reference has no sanitizer configuration (CMakeLists.txt:48-66)."""
    build_hip(False, verbose)
    obj = os.path.join(outdir, "obj")
    for d in (obj, os.path.join(outdir, "lib"), os.path.join(outdir, "bin")):
        os.makedirs(d, exist_ok=True)
    rt = os.path.join(CSRC, "runtime")
    cpp = os.path.join(CSRC, "cpp")
    inc = os.path.join(cpp, "include")
    hd = os.path.join(CSRC, "hip")
    _embed_opencl_sources(os.path.join(cpp, "src", "core", "opencl"))
    base = ["g++", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-parameter"] + \
        SANITIZE_FLAGS
    # ---- _rt
    rt_srcs = [os.path.join(rt, f) for f in sorted(os.listdir(rt)) if f.endswith(".cc")]
    rt_srcs.append(os.path.join(CSRC, "bindings", "rt_bind.cc"))
    jobs, rt_objs = [], []
    for s in rt_srcs:
        o = os.path.join(obj, "rt_" + os.path.basename(s) + ".o")
        rt_objs.append(o)
        if _newer(o, [s] + _headers(rt)):
            jobs.append(base + ["-fvisibility=hidden", "-c", s, "-o", o] + _py_includes())
    # ---- C++ API
    flags = base + ["-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include"),
                    "-I" + inc]
    srcs = []
    for root, _, files in os.walk(os.path.join(cpp, "src")):
        srcs += [os.path.join(root, f) for f in sorted(files) if f.endswith(".cc")]
    srcs += [os.path.join(rt, f) for f in ("graph.cc", "logging.cc", "snapshot.cc")]
    hdrs = _headers(inc, rt, hd)
    cpp_objs = []
    for s in srcs:
        o = os.path.join(obj, "cpp_" + os.path.relpath(s, CSRC).replace(os.sep, "_") + ".o")
        cpp_objs.append(o)
        if _newer(o, [s] + hdrs):
            jobs.append(flags + ["-c", s, "-o", o])
    if jobs and verbose:
        print("[singa_amd.build] compiling %d sources with %s" % (
            len(jobs), SANITIZE_FLAGS[0]))
    _compile_many(jobs, 8)
    target = os.path.join(outdir, "_rt" + EXT_SUFFIX)
    _run(["g++", "-shared", "-o", target] + rt_objs + SANITIZE_FLAGS[:1] + ["-lpthread"])
    hip_objs = [os.path.join(OBJ, "hip_" + f + ".o")
                for f in sorted(os.listdir(hd)) if f.endswith(".hip")]
    lib = os.path.join(outdir, "lib", "libsinga_amd_cpp.so")
    _run(["g++", "-shared", "-o", lib] + cpp_objs + hip_objs + SANITIZE_FLAGS[:1] +
         ["-L" + os.path.join(ROCM, "lib"), "-lamdhip64",
          "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-lpthread"])
    exe = os.path.join(outdir, "bin", "singa_cpp_test")
    _run(flags + [os.path.join(cpp, "test", "test_main.cc"), "-o", exe,
                  "-L" + os.path.join(outdir, "lib"), "-lsinga_amd_cpp",
                  "-Wl,-rpath,$ORIGIN/../lib",
                  "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-lpthread"])
    return [target, lib, exe]


def build_all(force=False, verbose=True):
    return [build_rt(force, verbose), build_hip(force, verbose)] + \
        build_cpp(force, verbose)


if __name__ == "__main__":
    if "--sanitize" in sys.argv:
        outs = build_sanitized(sys.argv[sys.argv.index("--sanitize") + 1])
    else:
        outs = build_all(force="--force" in sys.argv)
    for o in outs:
        print("built", o)
