"""oracle/features_ref.py against the live reference's log_specgram (tests/golden/specgram.npz, oracle/gen_golden.py)."""
import os

import numpy as np

from oracle import features_ref


def test_matches_reference_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "specgram.npz"))
    feats = features_ref.log_specgram(z["audio"], int(z["sample_rate"]))
    assert feats.shape == z["feats"].shape == (101, 161)
    # the reference computes in float32 (scipy on int16 input); compare where the power is above float32 noise
    strong = z["feats"] > z["feats"].max() - 25.0
    np.testing.assert_allclose(feats[strong], z["feats"][strong], rtol=0, atol=2e-3)
    assert np.abs(feats - z["feats"]).mean() < 5e-3


def test_frame_count_of_the_reference_wav_fixtures():
    # /root/reference/tests: test0.wav has 17,622 samples -> (109, 161); test1.wav 25,130 -> (156, 161)
    for n, frames in ((17622, 109), (25130, 156)):
        assert features_ref.log_specgram(np.zeros(n), 16000).shape == (frames, 161)
