"""Builds the native library (HIP kernels + C-ABI) for gfx950 in-tree with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "libJpegXLFileTypeIO_X64.so"
TEST_LIBNAME = "libjxlhip_selftest.so"   # the product's objects + csrc/selftest.cc: writer self tests for the CPU suite, never shipped
TEST_SOURCES = ["selftest.cc"]
SOURCES = ["kernels.hip", "entropy_kernels.hip", "tile_kernels.hip", "encode_kernels.hip", "host_parse.cc", "host_write.cc", "icc.cc", "decoder.cc", "encoder.cc"]
HEADERS = ["dev_types.h", "dev_util.h", "modular_uniform.h", "enc_types.h", "kernels.h", "host_parse.h", "host_write.h", "icc.h", os.path.join("..", "..", "include", "jxlfiletypeio.h")]


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


def test_lib_path():
    return os.path.join(LIBDIR, TEST_LIBNAME)


def source_digest():
    """SHA-256 over the sources and headers the library is built from.  build() records it next to the library; api.lib() compares it,
    because the library is git-ignored yet travels to the GPU box: after an edit a suite could run green against the old binary
    (file times do not survive the copy, contents do)."""
    import hashlib
    h = hashlib.sha256()
    for name in SOURCES + TEST_SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def manifest_text(extra=None):
    """What build() records and api.lib() expects: the source digest, plus the extra compiler flags of a non-default build (a library
    built with -DJXLHIP_EXPERIMENTS / -DJXLHIP_PROFILE_HF loads only in a process that asks for those flags too)."""
    if extra is None:
        extra = os.environ.get("JXLHIP_EXTRA_CFLAGS", "")
    extra = " ".join(extra.split())
    return source_digest() + ("\nextra: " + extra if extra else "") + "\n"


def manifest_path():
    return os.path.join(LIBDIR, "sources.sha256")


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    test_objs = []
    procs = []
    extra = os.environ.get("JXLHIP_EXTRA_CFLAGS", "")
    for src in SOURCES + TEST_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src + ".o")
        (test_objs if src in TEST_SOURCES else objs).append(o)
        # an object built with other extra flags (e.g. -DJXLHIP_PROFILE_HF) is stale even when its sources have not changed
        flagfile = o + ".flags"
        old_extra = open(flagfile).read() if os.path.exists(flagfile) else ""
        if force or extra != old_extra or _stale(o, [s] + hdrs):
            # the old object and its flag record go first: a failed compile must not leave an object that the next build takes
            # for one built with the new flags (the record is written only after hipcc has returned 0)
            for stale in (o, flagfile):
                if os.path.exists(stale):
                    os.remove(stale)
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-x", "hip", "-c", s, "-o", o] + extra.split()
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, flagfile, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = []
    for src, flagfile, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append("hipcc failed for %s:\n%s" % (src, out.decode(errors="replace")))
            continue
        with open(flagfile, "w") as ff:
            ff.write(extra)
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    if failed:
        raise RuntimeError("\n".join(failed))
    out = lib_path()
    test_out = test_lib_path()
    if force or procs or _stale(test_out, objs + test_objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", test_out] + objs + test_objs + ["-lpthread"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link of the test library failed:\n" + r.stdout.decode(errors="replace"))
    if force or procs or _stale(out, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-lpthread"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
        # the name the reference's P/Invoke asks for (src/Interop/JpegXL_X64.cs:19)
        alias = os.path.join(LIBDIR, "JpegXLFileTypeIO_X64.dll")
        if os.path.islink(alias) or os.path.exists(alias):
            os.remove(alias)
        os.symlink(LIBNAME, alias)
    with open(manifest_path(), "w") as f:
        f.write(manifest_text(extra))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
