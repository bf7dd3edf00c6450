"""Channel schedule of the encoder-decoder (mirrors reference nlt/util/net.py:18-56)."""
import math


def gen_feat_n(min_n, max_n, final_n=3):
    """Channel counts after the first (original-resolution) layer, e.g.
    (16, 256) -> [16, 32, 64, 128, 256, 256, 128, 64, 32, 16, 8, 4, 3].

    Doubling from `min_n` up to `max_n`, mirrored back down, then halving towards `final_n`
    (never below it) and finally `final_n` itself -- same sequence as the reference for
    every input it accepts."""
    if not (max_n >= min_n and max_n >= final_n):
        raise AssertionError("Max number of channels must be >= the min and the final number of channels")
    up = [2 ** e for e in range(int(math.log2(min_n)) + 1, int(math.log2(max_n)) + 1)]
    if not up or up[0] != min_n:
        up.insert(0, min_n)
    if up[-1] != max_n:
        up.append(max_n)
    seq = up + up[::-1]
    e = int(math.log2(seq[-1])) - 1
    while e > int(math.log2(final_n)):
        seq.append(2 ** e)
        e -= 1
    while seq and seq[-1] < final_n:
        seq.pop()
    seq.append(final_n)
    return seq
