"""Worker of tests/test_gpu_parity.py::test_all_kernel_variants_agree: kernel variants x launch shapes against the oracle, in a
process of its own so that it can run against the tuning build of the library (SLICESLICE_HIP_LIB)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sliceslice_rs_amd as ss  # noqa: E402
from oracle import oracle as O  # noqa: E402

SEED_NEEDLE = 0x5EED0002


def absent_needle(n):
    nd = bytearray(ss.fill_random_host(n, SEED_NEEDLE).tobytes())
    nd[0 if n == 1 else (1 if n == 2 else n // 2)] = 0xFF
    return bytes(nd)


def main():
    which = sys.argv[1]
    tuning = ss.lib().has_hooks and b"tuning" in ss.lib().ss_version()
    assert tuning == (which == "tuning"), which
    ln = (8 << 20) + 777
    t = torch.empty(ln + 16, dtype=torch.uint8, device="cuda")
    ss.fill_random_device(t, 0xABCDEF)
    t = t[3:3 + ln]
    host = t.cpu().numpy()
    cases = [absent_needle(n) for n in (1, 2, 16, 20, 200, 700, 1200)]
    cases += [host[ln - n:].tobytes() for n in (1, 2, 16, 20, 200, 700, 1200)]
    cases += [host[12345:12345 + n].tobytes() for n in (33, 100, 257, 1000)]
    # variant = 100000*B + 10000*OCC + 1000*LAYOUT + 100*MODE + 10*U + NT (include/sliceslice_hip_tuning.h), tuning build only:
    # launch-shape digits (1xxxxx / 3xxxxx = 128- / 512-thread workgroups, x4xxxx = occupancy cap), plain loads, U = 8, the 8-byte
    # phase for two-byte filters (2xxx) and the 16-byte layout for one-byte needles (1xxx), the cross-lane kernels without the third
    # byte (x3xx: what the census selects for pairs that rarely match).  The product library launches the
    # automatic choice only (variant 0, grid 0) and has no entry point to ask for anything else.
    variants = (41, 100041, 300041, 40041, 40, 80, 81, 240, 241, 280, 281, 340, 341, 381, 1040, 1041, 1081, 2040, 2041, 2080, 2081, 100241,
                100341, 302041, 130081) if tuning else (0,)
    checked = refused = 0
    for nd in cases:
        want = O.OracleSearcher(nd).search_in(host)
        for variant in variants:
            for grid in ((0, 1, 7, 4096, -1, -3, -1000) if tuning else (0,)):
                s = ss.DynamicHipSearcher.new(nd)
                s.set_variant(variant)
                s.set_grid(grid)
                assert s.search_in(t) == want, (len(nd), variant, grid)
                checked += 1
            if len(nd) >= 16 and variant in (2040, 2041, 2080, 2081, 302041):
                # the 8-byte layout's THREE-byte first phase: all three filter bytes within four bytes (filter_half3_near)
                for a, b, c in ((0, 1, 2), (0, 3, 2), (5, 8, 6), (len(nd) - 4, len(nd) - 1, len(nd) - 3)):
                    s = ss.DynamicHipSearcher.new(nd)
                    s.set_filter(a, b, c)
                    s.set_variant(variant)
                    assert s.search_in(t) == want, (len(nd), variant, a, b, c)
                    checked += 1
            if len(nd) > 16:
                # the reference's pair (needle[0], needle[n-1]), which no constructor picks at this distance: cross-lane kernels up
                # to a distance of 1,007, beyond that the first byte + two partners with the far byte checked in memory
                for grid in ((0, 7, -3) if tuning else (0,)):
                    s = ss.DynamicHipSearcher.new(nd)
                    s.set_filter(0, len(nd) - 1)
                    assert s.filter3 == (0, len(nd) - 1, len(nd) - 1)
                    s.set_variant(variant)
                    s.set_grid(grid)
                    assert s.search_in(t) == want, (len(nd), variant, grid, "reference pair")
                    if (variant // 10) % 10 != 8:                      # find() has the U = 4 kernels only
                        assert s.find(t) == (host.tobytes().find(nd) if want else None), (len(nd), variant, grid, "reference pair, find")
                    checked += 1
        if not tuning:
            # the product library has no variant / grid override at all - asking is refused loudly, nothing is launched
            s = ss.DynamicHipSearcher.new(nd)
            for ask in (lambda: s.set_variant(81), lambda: s.set_grid(7)):
                try:
                    ask()
                    raise AssertionError("the product library took a tuning override")
                except ss.SlicesliceError as e:
                    assert e.code == ss.SS_ERR_ARGUMENT and "SS_TEST_HOOKS" in str(e), e
                    refused += 1
            assert s.search_in(t) == want
    print("variants ok: %d searches checked, %d refusals" % (checked, refused))


if __name__ == "__main__":
    main()
