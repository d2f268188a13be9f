#!/usr/bin/env python3
"""the CPU side, timed on THIS host with the C/AVX2 restatement of the reference
path (NOT the Rust binary, which cannot be built in this image).  The measurement itself lives in bench.py's
CPU-baseline leg (`bench.py --cpu-report`, no GPU needed); this is a convenience wrapper that prints its JSON
lines: config 1 long + short (beside README's 35.181 / 79.416 ms), the synthetic 1 GiB haystack at 1 thread
and all threads, and the needle-length sweep {1,2,4,8,16,32,128} at 1 thread."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-report"]))
