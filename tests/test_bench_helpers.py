"""bench.py's bookkeeping that needs no GPU: the per-round profile lookup and the staleness stamp of the replayed
HBM-traffic figure (a profile taken on other kernel sources must be visible as stale)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_latest_profile_picks_the_newest_round_and_names_its_file():
    import bench
    tj = bench.latest_profile("traffic_dominant_kernel")
    assert tj is not None and tj["_file"].startswith("profiles/r") and tj["_file"].endswith("_traffic_dominant_kernel.json")
    rounds = sorted(f[:3] for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic_dominant_kernel.json"))
    assert os.path.basename(tj["_file"]).startswith(rounds[-1])
    for key in ("kernel", "traffic_bytes_per_launch", "commit", "kernel_source_sha256", "command"):
        assert key in tj, key
    assert bench.latest_profile("no_such_profile") is None


def test_kernel_source_hash_tracks_the_convolution_sources(tmp_path):
    import bench
    h = bench.kernel_source_sha256()
    assert len(h) == 64 and h == bench.kernel_source_sha256()
    # the committed traffic profile must have been taken on the sources in the tree (re-take the PMC passes after
    # touching csrc/conv.hip, common.h or the Makefile: tools/pmc_traffic.py)
    tj = bench.latest_profile("traffic_dominant_kernel")
    if tj["kernel_source_sha256"] != h:
        import pytest
        pytest.skip("profiles/*_traffic_dominant_kernel.json was taken on other kernel sources: bench.py reports "
                    "kernel_source_unchanged_since = false until the --pmc passes are re-taken (tools/pmc_traffic.py)")
