"""CPU: the C-ABI shared library loads and exports every symbol include/vacnic_hip.h declares (and the ctypes
binding covers exactly that set).  No compute calls here (no GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "vacnic_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vacnic_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built_lib):
    lib = ctypes.CDLL(built_lib)
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vacnic_hip.h but not exported by libvacnic_hip.so"


def test_binding_covers_header(built_lib):
    from vacnic_amd import _lib
    assert sorted(_lib.EXPORTED) == declared_functions()
    assert _lib.lib.vacnic_version() >= 100
    assert _lib.lib.vacnic_last_error_string() is not None


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """every argument struct of the binding has the size and the field offsets the C compiler gives the struct of the same name in
    include/vacnic_hip.h (a field added on one side only, or in another order, would silently shift every later argument)."""
    import shutil
    import subprocess
    from vacnic_amd import _lib
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no C compiler")
    structs = {n: t for n, t in vars(_lib).items() if isinstance(t, type) and issubclass(t, ctypes.Structure) and t is not ctypes.Structure}
    structs = {t.__name__: t for t in structs.values() if t.__name__.startswith("vacnic_")}
    assert len(structs) >= 15, sorted(structs)
    header = open(os.path.join(ROOT, "include", "vacnic_hip.h")).read()

    def member(fname):            # the binding flattens small arrays: seq0 / seq1 <-> seq[2]
        m = re.match(r"^(\w+?)(\d+)$", fname)
        return f"{m.group(1)}[{m.group(2)}]" if m and re.search(rf"\b{m.group(1)}\[\d+\]", header) else fname
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vacnic_hip.h"', "int main(void) {"]
    for name, t in sorted(structs.items()):
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for fname, *_ in t._fields_:
            lines.append(f'  printf("{name}.{fname} %zu\\n", offsetof({name}, {member(fname)}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True, capture_output=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, t in structs.items():
        assert int(got[name]) == ctypes.sizeof(t), (name, got[name], ctypes.sizeof(t))
        for fname, *_ in t._fields_:
            assert int(got[f"{name}.{fname}"]) == getattr(t, fname).offset, (name, fname)


def test_bad_arguments_return_status_not_abort(built_lib):
    """error convention: status code + message, never an abort (validation happens before any launch)."""
    import pytest
    from vacnic_amd import _lib
    with pytest.raises(ValueError, match="null operand"):
        _lib.call_struct("vacnic_gemm_bf16", stream=None, x=None, w=None, bias=None, out=None, preact=None, dact_src=None,
                         residual=None, M=1, N=1, K=8, ldx=8, ldw=8, ldo=1, x_kstrided=0, w_kstrided=0, act=0, out_mode=0,
                         split_k=1, alpha=1.0)
    with pytest.raises(ValueError, match="multiples of 8"):
        _lib.call_struct("vacnic_gemm_bf16", stream=None, x=16, w=16, bias=None, out=16, preact=None, dact_src=None,
                         residual=None, M=4, N=4, K=8, ldx=9, ldw=8, ldo=4, x_kstrided=0, w_kstrided=0, act=0, out_mode=0,
                         split_k=1, alpha=1.0)
    with pytest.raises(ValueError, match="must be a multiple of 8"):
        _lib.call_struct("vacnic_add_ln_fwd", stream=None, x=16, residual=None, gamma=16, beta=16, out=16, mean=None, rstd=None,
                         R=4, D=100, eps=1e-5, p_drop=0.0, seed=0)


def test_product_path_never_imports_oracle():
    import glob
    for f in glob.glob(os.path.join(ROOT, "vacnic_amd", "**", "*.py"), recursive=True):
        src = open(f).read()
        if f.endswith("smoke.py"):
            continue        # __graft_entry__.smoke(): the oracle is the checker there
        assert "oracle" not in src.replace("the oracle", "").replace("oracle's", ""), f"{f} must not reference oracle/"
