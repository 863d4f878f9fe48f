"""Chained multi-frame parity: 30 frames of a synthetic stereo sequence through the tracking front-end of
tests/tracking_chain.py - the state (map points, holders, observation counts, temporal points) carried from frame to
frame - by the CPU oracle and by the HIP library (host-array entry points, the device-resident *_device entry points, and the latency path
orbx_stereo_frame_view + device-resident matchers).
Every frame's keypoints, descriptors, mvuRight, mvDepth, SearchByProjection holders, isInFrustum records, SearchLocalPoints
holders and the map-point assignment must be identical: a single differing bit anywhere diverges the chains for good."""
import importlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tracking_chain as tc   # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,nf,T", [(1241, 376, 2000, 30), (752, 480, 1000, 12)], ids=["kitti_2000feat_30frames", "euroc_12frames"])
def test_chained_tracking_front_end(oracle, w, h, nf, T):
    synth = importlib.import_module(tc.PKG + ".synth")
    step = 0.04
    frames, _ = synth.stereo_sequence(w, h, T, k=11, step=step)
    Ts = tc.poses(T, step)
    chains = [tc.Chain(B(w, h, nf), w, h, nf) for B in (tc.OracleBackend, tc.GpuHostBackend, tc.GpuStereoFrameBackend, tc.GpuDeviceBackend, tc.GpuViewBackend)]
    for t in range(T):
        snaps = [c.step(frames[t][0], frames[t][1], Ts[t]) for c in chains]
        for c in chains[1:]:
            diff = tc.first_difference(chains[0].log, c.log)
            assert diff is None, "%s diverges from the oracle at frame %d, field %s" % (c.be.name, diff[0], diff[1])
    log = chains[0].log
    # the sequence really exercises the chain: projection matches every frame, local-map matches, outliers, a growing map
    assert min(s["proj_n"] for s in log[1:]) >= 20
    assert sum(s["local_n"] for s in log[1:]) > 50
    assert log[-1]["map_size"] > log[0]["map_size"]
    assert sum(s["outliers"] for s in log[1:]) > 0
