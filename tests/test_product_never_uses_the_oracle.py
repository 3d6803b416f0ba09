"""Static guard (CPU): oracle/ is test infrastructure.  No product source — the package gs-sdf_amd/ (Python, C++, HIP, its Makefiles) and
include/ — may import, include, link, load or execute anything under oracle/ (the CPU restatement, the compiled reference in oracle/_ref);
bench.py may only inside its cpu_baseline leg; and the product must fail loudly when the HIP library is missing instead of falling back."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = re.compile(r"(^\s*(from|import)\s+oracle\b|import_module\(\s*[\"']oracle|[\"'/]oracle/|liborc_|_gsdf_reference|oracle\._ref|oracle\.ref_link)")


def _sources(base, exts):
    for d, dirs, files in os.walk(base):
        dirs[:] = [x for x in dirs if x not in ("_obj", "__pycache__", "lib")]
        for f in files:
            if f.endswith(exts) or f == "Makefile":
                yield os.path.join(d, f)


def _strip_comments(path, text):
    if path.endswith(".py"):
        text = re.sub(r'"""(?:.|\n)*?"""', "", text)
        return "\n".join(l.split("#", 1)[0] for l in text.splitlines())
    text = re.sub(r"/\*(?:.|\n)*?\*/", "", text)
    return "\n".join(l.split("//", 1)[0] for l in text.splitlines())


def test_no_product_source_touches_the_oracle():
    bad = []
    for base in (os.path.join(ROOT, "gs-sdf_amd"), os.path.join(ROOT, "include")):
        for p in _sources(base, (".py", ".cpp", ".h", ".hip", ".hpp")):
            code = _strip_comments(p, open(p, errors="replace").read())
            for n, line in enumerate(code.splitlines(), 1):
                if CODE.search(line):
                    bad.append(f"{os.path.relpath(p, ROOT)}:{n}: {line.strip()[:120]}")
    assert not bad, "product sources reference oracle/:\n" + "\n".join(bad)


def test_bench_uses_the_oracle_only_in_its_cpu_baseline_leg():
    """bench.py and benchlib/: the oracle is imported by benchlib/cpu_baseline.py alone (the CPU baseline + parity leg)"""
    paths = [os.path.join(ROOT, "bench.py")] + sorted(os.path.join(ROOT, "benchlib", f) for f in os.listdir(os.path.join(ROOT, "benchlib")) if f.endswith(".py"))
    for p in paths:
        src = _strip_comments(p, open(p).read())
        uses = [m.group(0).strip() for m in re.finditer(r"^\s*(from oracle\b.*|import oracle\b.*)$", src, re.M)]
        if os.path.basename(p) == "cpu_baseline.py":
            assert uses, "the cpu_baseline leg no longer times the oracle?"
        else:
            assert not uses, f"{os.path.relpath(p, ROOT)} imports the oracle: {uses}"
        assert "_gsdf_reference" not in src and "ref_link" not in src


def test_the_product_fails_loudly_without_its_library(monkeypatch, tmp_path):
    import pytest
    import torch
    import gs_sdf_amd.capi as capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "libgsdf_hip.so"))
    with pytest.raises(RuntimeError, match="not found"):
        capi.lib()
    import gs_sdf_amd.ops as ops
    with pytest.raises(RuntimeError):                       # an operator call does not quietly compute something else
        ops.distCUDA2(torch.rand(16, 3))
    monkeypatch.undo()
    with pytest.raises(RuntimeError, match="device tensor"):   # and host tensors are refused at the boundary
        capi.ptr(torch.zeros(4), name="x")
