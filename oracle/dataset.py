"""Oracle restatement of the dataset window / style-example index rules (numpy ints).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Reference: ZEGGS/dataset.py:79-96 (window table), :110-153 (__getitem__),
:176-204 (get_example incl. tail-repeat padding); ZEGGS/helpers.py:26-37
(split_by_ratio, used by "stitch" blending in generate.py:280-298).
"""
import numpy as np


def build_windows(ranges_train, window):
    """dataset.py:79-96 -> (start[Nwin] int64, sample[Nwin] int16).
    Window i covers frames start[i] .. start[i]+window-1."""
    starts, samples = [], []
    for s, (a, b) in enumerate(ranges_train):
        for ri in range(int(a), int(b) - window):
            starts.append(ri)
            samples.append(s)
    return np.asarray(starts, dtype=np.int64), np.asarray(samples, dtype=np.int16)


def example_range(r0, window, range_start, range_end, example_len, n_total):
    """dataset.py:180-187 -> (start, end) frame indices of the style example."""
    r_last = r0 + window - 1
    ext = (example_len - window) // 2
    ws = min(ext, r0 - range_start)
    we = min(ext, range_end - r_last)
    s_ext = ws + ext - we
    w_ext = we + ext - ws
    start = max(r0 - s_ext, range_start)
    end = min(r_last + w_ext, range_end) + 1
    end = min(end, n_total)
    return int(start), int(end)


def example_rows(start, end, example_len):
    """dataset.py:198-203: source frame index of each of the example_len rows
    (short examples are padded by repeating their tail)."""
    cur = end - start
    rows = list(range(start, end))
    if cur < example_len:
        need = example_len - cur
        rows += list(range(end - need, end))      # example[-need:]
    return np.asarray(rows, dtype=np.int64)


def split_by_ratio(n, ratios):
    """helpers.py:26-37 -> list of (start, end) integer frame splits."""
    assert sum(ratios) == 1.0
    out, prev_end = [], 0
    for r in ratios:
        s = int(prev_end)
        e = int(prev_end + r * n)                 # truncation, not rounding
        out.append([s, e])
        prev_end = e
    out[-1][1] = n
    return [tuple(x) for x in out]
