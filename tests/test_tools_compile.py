"""The measurement scripts under tools/ only run on the GPU box; keep them at least syntactically alive here."""
import glob
import os
import py_compile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))


@pytest.mark.parametrize("path", TOOLS, ids=[os.path.basename(p) for p in TOOLS])
def test_tool_compiles(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "out.pyc"), doraise=True)


def test_tools_do_not_read_the_reference_tree():
    # nothing that runs on the GPU box may depend on /root/reference (it does not exist there)
    for path in TOOLS + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        with open(path) as f:
            src = f.read()
        assert "/root/reference" not in src, path
