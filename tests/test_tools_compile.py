"""Every script under tools/ (profiling, probes, one-off parity runs: they only ever execute on the GPU box) at least parses here."""
import glob
import os
import py_compile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))), ids=os.path.basename)
def test_tool_script_compiles(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)
