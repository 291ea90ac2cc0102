"""The shell and python helpers under tools/ at least parse (they only run on the GPU box), and the results table prints from the
committed profile round."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shell_tools_parse():
    for f in sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh"))):
        r = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stderr)


def test_python_tools_compile():
    for f in sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))):
        compile(open(f).read(), f, "exec")


def test_probe_lists_have_three_fields():
    for f in sorted(glob.glob(os.path.join(ROOT, "tools", "lists", "*.txt"))):
        for line in open(f):
            if line.strip():
                assert line.count("|") == 2, (f, line)


def test_results_table_prints_the_committed_round():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "results_table.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "MIN_DISTANCE exact (headline)" in r.stdout and "roofline:" in r.stdout
