class AdaptiveImageLossFunction:
    def __init__(self, *a, **kw):
        raise NotImplementedError("tests/tf_shim does not run robust_loss (pinned by the reference's own golden files instead)")
