"""Output-directory bookkeeping for training scripts (reference
pfrl/experiments/prepare_output_dir.py:67-160): creates ``<basedir>/<exp_id>`` and
records how the run was started.

Files written: ``start.txt`` (one timestamp per (re)start, appended), ``args.txt``
(JSON of the arguments), ``environ.txt`` (JSON of the environment), ``command.txt``
(the command line) and, under git, ``git-head.txt`` / ``git-status.txt`` /
``git-log.txt`` / ``git-diff.txt``.  Without an explicit ``exp_id`` the id is
``[prefix-]<HEAD sha>-<crc32(git diff HEAD)>-<crc32(pickled argv)>`` under git and
the timestamp otherwise; an existing directory is first copied to
``<outdir>.<timestamp>.backup``.
"""
import argparse
import datetime
import json
import os
import pickle
import shutil
import subprocess
import sys
from binascii import crc32

_GIT_RECORDS = (("git-head.txt", "git rev-parse HEAD"), ("git-status.txt", "git status"),
                ("git-log.txt", "git log"), ("git-diff.txt", "git diff HEAD"))


def _git(cmd):
    return subprocess.check_output(cmd.split())


def is_under_git_control():
    with open(os.devnull, "wb") as null:
        try:
            return subprocess.call(["git", "rev-parse"], stdout=null, stderr=null) == 0
        except OSError:
            return False


def generate_exp_id(prefix=None, argv=sys.argv):
    """Deterministic id from the git state and the command line."""
    if not is_under_git_control():
        raise RuntimeError("Cannot generate experiment id due to Git lacking.")
    parts = [] if prefix is None else [prefix]
    parts.append(_git("git rev-parse HEAD").strip().decode())
    parts += ["%08x" % crc32(blob) for blob in (_git("git diff HEAD"), pickle.dumps(argv))]
    return "-".join(parts)


def save_git_information(outdir):
    for name, cmd in _GIT_RECORDS:
        with open(os.path.join(outdir, name), "wb") as f:
            f.write(_git(cmd))


def prepare_output_dir(args, basedir=None, exp_id=None, argv=None,
                       time_format="%Y%m%dT%H%M%S.%f", make_backup=True):
    now = datetime.datetime.now().strftime(time_format)
    under_git = is_under_git_control()
    if exp_id is None:
        exp_id = generate_exp_id() if under_git else now
    outdir = os.path.join(basedir or ".", exp_id)
    if make_backup and os.path.exists(outdir):
        shutil.copytree(outdir, "{}.{}.backup".format(outdir, now))
    os.makedirs(outdir, exist_ok=True)

    def write(name, text, mode="w"):
        with open(os.path.join(outdir, name), mode) as f:
            f.write(text)

    write("start.txt", datetime.datetime.now().strftime("%Y%m%dT%H%M%S.%f") + "\n", mode="a")
    write("args.txt", json.dumps(vars(args) if isinstance(args, argparse.Namespace) else args))
    write("environ.txt", json.dumps(dict(os.environ)))
    write("command.txt", " ".join(sys.argv if argv is None else argv))
    if under_git:
        save_git_information(outdir)
    return outdir
