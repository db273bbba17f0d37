"""Build recipe of ``oracle/_ref/``: the REFERENCE ITSELF, compiled for the GPU box.

pfnet/pfrl is pure Python.  Its sources are never copied into this repository; what this recipe
writes is the interpreter's compiled form of them -- sourceless ``.pyc`` files, the Python
counterpart of the ``.so`` a C reference would be built into -- from the sources where they lie
under ``/root/reference``, outputs only into ``oracle/_ref/`` (git-ignored, but it travels to
the GPU box with the rest of the tree, like the built ``.so`` files).  CPython imports a
directory of ``.pyc`` files like a package, provided the interpreter version matches (both
sides run this image's Python 3.10).

Used by: ``bench.py``'s ``cpu_baseline`` leg (tools/reference_cpu_baseline.py: the reference's own
train loop, ``gpu=-1``, timed on the GPU box's host cores -> ``kind: "reference"``) and by the
test that runs the reference's Atari example script against this package
(tests/test_reference_examples.py).  TEST / MEASUREMENT INFRASTRUCTURE ONLY: nothing under
``pfrl_amd/`` may import it (tests/test_host_logic.py checks).

    python oracle/build_ref.py            # no-op with a message when /root/reference is absent
"""
import hashlib
import json
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REFERENCE = os.environ.get("PFRL_REFERENCE", "/root/reference")
TREES = ("pfrl", "examples")


def build(force=False, quiet=False):
    """Compile the reference's packages into oracle/_ref/.  Returns the output directory, or None
    when the reference checkout is not present (the GPU box: it uses what was built here)."""
    if not os.path.isdir(os.path.join(REFERENCE, "pfrl")):
        if not quiet:
            print("oracle/build_ref.py: %s absent; keeping %s as it is" % (REFERENCE, OUT))
        return OUT if os.path.isdir(os.path.join(OUT, "pfrl")) else None
    files = []
    for tree in TREES:
        for root, dirs, names in os.walk(os.path.join(REFERENCE, tree)):
            dirs[:] = sorted(d for d in dirs if d != "__pycache__")
            for n in sorted(names):
                if n.endswith(".py"):
                    files.append(os.path.join(root, n))
    digest = hashlib.sha256()
    for f in files:
        digest.update(os.path.relpath(f, REFERENCE).encode())
        digest.update(open(f, "rb").read())
    stamp = dict(python="%d.%d" % sys.version_info[:2], sha256=digest.hexdigest(), files=len(files))
    stamp_path = os.path.join(OUT, "MANIFEST.json")
    if not force and os.path.exists(stamp_path):
        try:
            if json.load(open(stamp_path)) == stamp:
                return OUT
        except Exception:
            pass
    shutil.rmtree(OUT, ignore_errors=True)
    for f in files:
        rel = os.path.relpath(f, REFERENCE)
        dst = os.path.join(OUT, rel + "c")                  # x.py -> x.pyc, same layout
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile: the path tracebacks show (the reference file, for file:line citations)
        py_compile.compile(f, cfile=dst, dfile=os.path.join("/root/reference", rel), doraise=True,
                           optimize=0)
    # the reference reads its version from pfrl/version.py only; nothing else to carry
    with open(stamp_path, "w") as fh:
        json.dump(stamp, fh)
    if not quiet:
        print("oracle/_ref: %d modules of %s compiled (no sources copied)" % (len(files), REFERENCE))
    return OUT


def available():
    return os.path.isdir(os.path.join(OUT, "pfrl"))


if __name__ == "__main__":
    build(force="--force" in sys.argv)
