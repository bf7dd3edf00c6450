"""TEST-ONLY: nlt/losses.py imports robust_loss.adaptive at module level (it needs TensorFlow + TF-Probability).  The
orchestration fixtures use `loss = l2`; the Barron loss is pinned separately by the reference's own wavelet_golden.mat and
partition_spline.npz (tests/test_oracle_barron.py).  See ../README.md."""
