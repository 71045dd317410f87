"""Every global name a function refers to must exist in its module (or be a builtin): catches a renamed or
out-of-scope variable in a branch the other tests do not run (no GPU needed: nothing is executed)."""
import builtins
import glob
import importlib.util
import os
import symtable

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, "gan_amd", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py")) +
               [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")])


def _unresolved(table, module_names, path, out):
    for child in table.get_children():
        if child.get_type() == "function":
            for sym in child.get_symbols():
                if sym.is_global() and sym.is_referenced() and not sym.is_assigned():
                    name = sym.get_name()
                    if name not in module_names and not hasattr(builtins, name):
                        out.append("%s: %s() refers to undefined global %r" % (os.path.relpath(path, ROOT), child.get_name(), name))
        _unresolved(child, module_names, path, out)


@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(f, ROOT) for f in FILES])
def test_functions_refer_only_to_names_their_module_defines(path):
    src = open(path).read()
    top = symtable.symtable(src, path, "exec")
    module_names = {s.get_name() for s in top.get_symbols()}       # assigned, imported or defined at module level
    out = []
    _unresolved(top, module_names, path, out)
    assert not out, "\n".join(out)
