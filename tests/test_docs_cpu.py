"""The documents cite files; every cited path under tools/, tests/, profiles/, oracle/, include/ or the package must exist - or be
named in tools/RETIRED.md, which says what the retired tool measured and in which commit it last lived (seven
citations pointed at scripts a clean-up had deleted, among them the only source of a set of published figures).  CPU only."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md"] + sorted(
    os.path.relpath(p, ROOT) for p in glob.glob(os.path.join(ROOT, "profiles", "*", "README.md")) + glob.glob(os.path.join(ROOT, "CHANGES_r*.md")))
CITED = re.compile(r"(?<![\w/.-])((?:tools|tests|profiles|oracle|include|sliceslice-rs_amd)/[A-Za-z0-9_./*-]+)")


@pytest.mark.parametrize("doc", DOCS)
def test_every_cited_path_exists_or_is_listed_as_retired(doc):
    retired = set(re.findall(r"`((?:tools|tests)/[A-Za-z0-9_./-]+)`", open(os.path.join(ROOT, "tools", "RETIRED.md")).read()))
    dead = []
    for m in CITED.finditer(open(os.path.join(ROOT, doc)).read()):
        p = m.group(1).split("::")[0].rstrip(".,:;)")
        if p.endswith(".rs") and not p.startswith("sliceslice-rs_amd/"):
            continue                                    # a file of the reference (tests/i386.rs, ...), cited relative to /root/reference
        if p.rstrip("/") in ("oracle/_ref",):
            continue                                    # the convention's name for a reference build this image cannot make (no rustc)
        if p.endswith((".so", ".o")) or "/libsliceslice" in p or p.endswith("native_bench") or (p.endswith("kernel_resources.json") and "csrc" in p):
            continue                                    # build products (git-ignored, made by build())
        ok = bool(glob.glob(os.path.join(ROOT, p))) if "*" in p else os.path.exists(os.path.join(ROOT, p))
        if not ok and p not in retired and p + ".hip" not in retired:        # (a retired probe may be cited by its binary's name)
            dead.append(p)
    assert not dead, "%s cites paths that do not exist and are not in tools/RETIRED.md: %s" % (doc, sorted(set(dead)))


def test_tools_import_only_modules_that_exist_and_the_evidence_files_hold_something():
    """A clean-up once deleted tools/settle.py while three tools still imported it: `tools/bench_configs.py` died at its first line on
    the GPU box and two rounds' `configs_1_3_5_text.jsonl` were committed EMPTY.  So: every local module a tool imports exists, and
    no tracked file under profiles/ is empty."""
    tools = os.path.join(ROOT, "tools")
    local = {os.path.splitext(f)[0] for f in os.listdir(tools) if f.endswith(".py")}
    stdlib_or_installed = {"argparse", "ctypes", "json", "os", "sys", "time", "random", "statistics", "subprocess", "csv", "glob", "shutil", "re",
                           "numpy", "torch", "sliceslice_rs_amd", "threading", "math", "struct", "collections", "itertools", "tempfile", "signal"}
    missing = []
    for f in sorted(os.listdir(tools)):
        if not f.endswith(".py"):
            continue
        for m in re.finditer(r"^\s*(?:from|import)\s+([A-Za-z_][A-Za-z0-9_]*)", open(os.path.join(tools, f)).read(), re.M):
            name = m.group(1)
            if name not in local and name not in stdlib_or_installed and name != "tools_settle":     # (a placeholder ab_compare.py rewrites)
                missing.append((f, name))
    assert not missing, "tools import modules that are neither local nor known: %s" % missing
    empty = [os.path.relpath(p, ROOT) for p in glob.glob(os.path.join(ROOT, "profiles", "**", "*"), recursive=True)
             if os.path.isfile(p) and os.path.getsize(p) == 0]
    assert not empty, "empty evidence files: %s" % empty
