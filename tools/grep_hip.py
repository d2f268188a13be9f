#!/usr/bin/env python3
"""./grep_hip.py <needle> <file> [--rare-position] - the reference's examples/grep.rs:42-56 with the "hip"
backend: map the file, build one searcher, one search_in, print the boolean."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliceslice_rs_amd as ss  # noqa: E402


def main():
    if len(sys.argv) < 3:
        raise SystemExit("./grep_hip.py <needle> <file>")
    needle, filename = sys.argv[1].encode(), sys.argv[2]
    searcher = ss.DynamicHipSearcher.new(needle)
    print("Searching for %s in %r: %s" % (sys.argv[1], filename, str(ss.search_file(searcher, filename)).lower()))


if __name__ == "__main__":
    main()
