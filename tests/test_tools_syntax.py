"""The scripts under tools/ and examples/ are run by hand on the GPU box; here they only have to parse
(python: py_compile, shell: bash -n), so that a rename in the package does not leave one of them behind unnoticed."""
import glob
import os
import py_compile
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) +
                                        glob.glob(os.path.join(ROOT, "examples", "*.py")) +
                                        [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]),
                         ids=os.path.basename)
def test_python_script_compiles(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "out.pyc"), doraise=True)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh"))), ids=os.path.basename)
def test_shell_script_parses(path):
    subprocess.run(["bash", "-n", path], check=True)


def test_tools_name_only_symbols_the_package_has():
    """every `from fastpm_amd import a, b, c` in tools/ and examples/ names things fastpm_amd exports"""
    import re
    import fastpm_amd
    for path in glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "examples", "*.py")):
        text = open(path).read()
        for m in re.finditer(r"from fastpm_amd import \(?([^)\n]+(?:\n[^)\n]+)*?)\)?\s*(?:#.*)?$", text, re.M):
            names = [n.strip() for n in m.group(1).replace("\n", " ").split(",") if n.strip()]
            for n in names:
                n = n.split("#")[0].strip().rstrip(")")
                if n and n.isidentifier():
                    assert hasattr(fastpm_amd, n), (os.path.basename(path), n)
