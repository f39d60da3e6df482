"""Build libsosgpu.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every csrc/*.hip is compiled to an object under build/ (in parallel, re-done only when the source or a header is
newer) and the objects are linked into libsosgpu.so next to this file."""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
SO = os.path.join(HERE, "libsosgpu.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value"]


def _headers():
    """What an object depends on besides its own source: every header AND every other source of csrc/ (sos_os_multi.hip and
    sos_stream_multi.hip #include sos_os.hip / sos_stream.hip), plus the C ABI header."""
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))] + \
           [os.path.join(HERE, "..", "include", "sosgpu.h")]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "sosgpu.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, objdir, force, verbose, extra):
    obj = os.path.join(objdir, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj):
        t = os.path.getmtime(obj)
        if all(os.path.getmtime(d) <= t for d in [path] + _headers()):
            return obj
    cmd = ["hipcc"] + FLAGS + list(extra) + ["-c", path, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return obj


def build(force=False, verbose=False, extra=(), out=None):
    """extra: additional hipcc flags (diagnostic builds, e.g. -DSOS_PROFILE_PHASES) -- such builds go to `out`
    with their own object directory."""
    out = out or SO
    if not force and not extra and out == SO and not needs_build():
        return SO
    objdir = OBJ if out == SO else OBJ + "_" + os.path.basename(out).replace(".", "_")
    os.makedirs(objdir, exist_ok=True)
    srcs = sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, objdir, force, verbose, extra), srcs))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
