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
