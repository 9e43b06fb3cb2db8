"""The in-tree binaries (git-ignored, shipped to the GPU box with the tree) must be the ones built from THIS tree:
metheor_amd/BUILD_INFO.json records the sha256 of every source they were built from; a stale .so fails here instead of
silently running old kernels ("I cannot say the driver recompiled")."""
import json
import os

import metheor_amd
from metheor_amd import build


def test_build_info_matches_the_tree():
    metheor_amd.lib()
    build.build()                                            # a no-op when everything is current
    info = json.load(open(build.BUILD_INFO))
    assert info["arch"] == "gfx950" and "-ffp-contract=off" in info["hip_flags"]
    assert info["sources"] == build.source_hashes(), "BUILD_INFO.json does not describe the sources in the tree"
    for name in ("libmetheor_hip.so", "libmetheor_host.so", "metheor"):
        p = os.path.join(os.path.dirname(build.__file__), name)
        assert os.path.exists(p) and info["artefacts"][name] == build._sha(p), name + " is not the file BUILD_INFO.json describes"
