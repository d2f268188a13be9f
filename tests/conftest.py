import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def checksums():
    with open(os.path.join(GOLDEN, "corpus_checksums.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def corpus():
    def rd(name):
        with open(os.path.join(GOLDEN, "data", name), "rb") as fh:
            return fh.read()
    words = rd("words.txt").split(b"\n")
    if words[-1] == b"":
        words.pop()
    return {"i386": rd("i386.txt"), "words": words, "haystack": rd("haystack"), "needle": rd("needle")}
