"""Build libxflow_amd.so (HIP kernels + host C++ + C ABI) and the xflow_lr CLI for gfx950.

In-tree, explicit hipcc: the .so travels to the GPU box with the snapshot.
  python -m xflow_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libxflow_amd.so")
CLI = os.path.join(LIBDIR, "xflow_lr")

LIB_SOURCES = ["xf_table.hip", "xf_model.hip", "xf_calib.hip", "xf_batch_dev.hip", "xf_cells_build.hip", "xf_cells_fwd.hip", "xf_cells_grad.hip", "xf_cells_grad_dense.hip", "xf_keybuild.hip", "xf_io.cc", "xf_batch.cc", "xf_metrics.cc",
               "xf_worker.cc", "xf_group.cc", "xf_modelfile.cc", "xf_sharded.hip", "xf_ingest.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result", "-Wno-unused-value",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


# experiments only (tools/grad_timeline.py: -DXF_GRAD_TIMELINE); part of the source hash, so a
# library built with extra flags never passes for the plain one
EXTRA = os.environ.get("XF_EXTRA_FLAGS", "").split()
FLAGS += EXTRA


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def source_hash():
    """sha256 over every source the library is built from (csrc/*, include/xflow_amd.h), in
    name order: compiled into the library (xf_source_hash) and checked when it is loaded, so
    that a stale prebuilt .so cannot stand in for newer sources unnoticed."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                   if f.endswith((".hip", ".cc", ".h")))
    files.append(os.path.join(ROOT, "include", "xflow_amd.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    if EXTRA:
        h.update(" ".join(EXTRA).encode())
    return h.hexdigest()[:32]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    deps = list(sources) + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(ROOT, "include", "xflow_amd.h"))
    return any(os.path.getmtime(d) > t for d in deps)


HASH_FILE = os.path.join(LIBDIR, "xf_source_hash.txt")


def _file_sha(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()[:32]


def _built_hash():
    """the source hash of the library on disk ("" when unknown).  Read from the sidecar file
    written when the library was linked — NOT by loading the library: glibc hands a second
    dlopen of the same path the handle it already has, so a probe that stays mapped would make
    the rebuilt file unloadable in this process (and would map /opt/rocm's HIP runtime before
    capi has preloaded torch's).  The sidecar names the library file it belongs to by content
    (file times do not survive a snapshot): one that names another file counts as unknown and
    forces a rebuild."""
    try:
        with open(HASH_FILE) as f:
            src, libsha = f.read().split()
        return src if libsha == _file_sha(LIB) else ""
    except (OSError, ValueError):
        return ""


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in LIB_SOURCES]
    if force or _stale(LIB, srcs) or _built_hash() != source_hash():
        objs = []
        procs = []
        stamp = os.path.join(LIBDIR, "xf_source_hash.cc")
        with open(stamp, "w") as f:   # (generated: the only place the hash is compiled in)
            f.write('extern "C" const char *xf_source_hash(void) { return "%s"; }\n'
                    % source_hash())
        for s in srcs + [stamp]:
            o = os.path.join(LIBDIR, os.path.basename(s) + ".o")
            objs.append(o)
            cmd = [_hipcc()] + FLAGS + (["-x", "hip"] if s.endswith(".hip") else []) + \
                  ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
        for cmd, p in procs:
            if p.wait() != 0:
                raise RuntimeError("build failed: " + " ".join(cmd))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(HASH_FILE, "w") as f:   # after the link: never older than the library
            f.write("%s %s\n" % (source_hash(), _file_sha(LIB)))
    cli_src = os.path.join(CSRC, "xf_cli.cc")
    if force or _stale(CLI, [cli_src, LIB]):
        cmd = [_hipcc()] + FLAGS + [cli_src, "-o", CLI, "-L" + LIBDIR, "-lxflow_amd",
                                    "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # the INTEGRATION.md binding (examples/ps_gpu.h) and its driver: a plain host program,
    # built with g++ against the C ABI only — what a maintainer of the reference would do
    demo_src = os.path.join(ROOT, "examples", "kv_demo.cc")
    demo = os.path.join(LIBDIR, "kv_demo")
    if os.path.exists(demo_src) and (force or _stale(
            demo, [demo_src, os.path.join(ROOT, "examples", "ps_gpu.h"), LIB])):
        cmd = ["g++", "-O2", "-std=c++11", "-Wall", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "examples"), demo_src, "-o", demo, "-L" + LIBDIR,
               "-lxflow_amd", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
