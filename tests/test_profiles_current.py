"""bench.py takes `roofline.traffic` (HBM bytes per launch, rocprofv3 PMC passes) from the profiles committed under profiles/.
A profile of an older build quotes kernels that no longer exist (round 4: profile_c3/c4 predated three kernel commits).  Every
profile of the newest round directory records the hash of the library sources it ran on (tools/profile_cfg.sh -> `kernels_sha`);
it must be this tree's."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from src_hash import src_hash  # noqa: E402

NEWEST = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*")))[-1]
PROFILES = sorted(glob.glob(os.path.join(NEWEST, "profile_c*.json")))


def test_the_newest_round_has_profiles_of_the_three_benched_configs():
    assert {os.path.basename(p) for p in PROFILES} >= {"profile_c2.json", "profile_c3.json", "profile_c4.json"}, NEWEST


@pytest.mark.parametrize("path", PROFILES, ids=[os.path.relpath(p, ROOT) for p in PROFILES])
def test_profile_was_taken_on_this_build(path):
    d = json.load(open(path))
    assert d.get("kernels_sha") == src_hash(), (f"{os.path.relpath(path, ROOT)} was taken on sources {d.get('kernels_sha')}, the tree is "
                                               f"{src_hash()}: run tools/profile_cfg.sh {d.get('config')} <tag> through gpurun and commit the result")
