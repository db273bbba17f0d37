"""``is_return_code_zero`` (reference pfrl/utils/is_return_code_zero.py): does a command succeed?"""
import subprocess


def is_return_code_zero(args):
    """True iff running ``args`` exits with status 0; its output is discarded, and a command that
    cannot be started counts as failure."""
    try:
        return subprocess.call(args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 0
    except OSError:
        return False
