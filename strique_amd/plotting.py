"""Figures of counted repeat expansions, drawn from the TSV contract of `count`.

The reference ships a `plot` command (scripts/STRique.py:948-1024) that shows, for every counted read, the raw
signal around the repeat and a close-up of each flank boundary.  This module is an independent take on the same
deliverable: everything it draws is derived from three TSV columns -- `offset` (first sample of the repeat),
`ticks` (its length in samples) and the two flank scores -- plus the read's raw signal from the fast5 index.

The geometry lives in `Windows` (pure integers, unit-tested without matplotlib); `draw` turns one read into a
figure whose shaded spans are exactly [offset, offset + ticks) in sample coordinates of the read.
"""
from collections import namedtuple

import numpy as np

CountRow = namedtuple("CountRow", "read_id target strand count score_prefix score_suffix log_p offset ticks")


def parse_counts(stream):
    """Rows of a `count` TSV (header and blank lines skipped, trailing columns such as `mod` ignored)."""
    for line in stream:
        cols = line.rstrip("\n").split("\t")
        if len(cols) < 9 or cols[0] == "ID":
            continue
        try:
            yield CountRow(cols[0], cols[1], cols[2], cols[3], float(cols[4]), float(cols[5]), float(cols[6]),
                           int(cols[7]), int(cols[8]))
        except ValueError:
            continue


class Windows(object):
    """Sample ranges of the three panels of one read, all half-open and clipped to [0, n).

    overview : the repeat with `extension` x ticks of context on either side
    left     : `zoom` samples either side of the first repeat sample (offset)
    right    : `zoom` samples either side of the first sample after the repeat (offset + ticks)
    repeat   : [offset, offset + ticks) itself
    """

    def __init__(self, n, offset, ticks, extension=0.1, zoom=500):
        clip = lambda v: int(min(max(v, 0), n))
        first, past = clip(offset), clip(offset + max(ticks, 0))
        margin = int(round(max(ticks, 0) * extension))
        self.n = n
        self.repeat = (first, past)
        self.overview = (clip(first - margin), clip(past + margin))
        self.left = (clip(first - zoom), clip(first + zoom))
        self.right = (clip(past - zoom), clip(past + zoom))

    def shaded(self, panel):
        """The part of the repeat that falls into `panel`, or None."""
        lo, hi = getattr(self, panel)
        a, b = max(lo, self.repeat[0]), min(hi, self.repeat[1])
        return (a, b) if b > a else None


def smooth(raw):
    """Three-sample running median (borders keep their own sample), then centre / scale robustly."""
    x = np.asarray(raw, np.float64)
    if x.size >= 3:
        mid = np.sort(np.stack([x[:-2], x[1:-1], x[2:]]), axis=0)[1]
        x = np.concatenate([x[:1], mid, x[-1:]])
    centre = np.median(x) if x.size else 0.0
    spread = np.median(np.abs(x - centre)) * 1.4826 if x.size else 1.0
    return (x - centre) / (spread if spread > 0 else 1.0)


PANELS = (("overview", "whole"), ("left", "begin"), ("right", "end"))


def draw(fig, raw, row, extension=0.1, zoom=500):
    """One read on `fig` (a matplotlib Figure).  Returns {panel: Axes}; every Axes carries `strique_span`, the
    (first, past) sample pair it shades, so that a caller or a test can read back what was drawn."""
    sig = smooth(raw)
    win = Windows(len(sig), row.offset, row.ticks, extension, zoom)
    grid = fig.add_gridspec(2, 2, height_ratios=(3, 2))
    axes = {"overview": fig.add_subplot(grid[0, :]), "left": fig.add_subplot(grid[1, 0]), "right": fig.add_subplot(grid[1, 1])}
    captions = {"overview": "%s  |  %s %s  |  %s repeat units  |  log p %.1f" % (row.read_id, row.target, row.strand, row.count, row.log_p),
                "left": "repeat begins at sample %d (prefix flank score %.3f)" % (win.repeat[0], row.score_prefix),
                "right": "repeat ends before sample %d (suffix flank score %.3f)" % (win.repeat[1], row.score_suffix)}
    for panel, _ in PANELS:
        ax = axes[panel]
        lo, hi = getattr(win, panel)
        t = np.arange(lo, hi)
        ax.plot(t, sig[lo:hi], color="0.25", linewidth=0.6 if panel == "overview" else 0.9)
        span = win.shaded(panel)
        ax.strique_span = span
        if span is not None:
            ax.axvspan(span[0], span[1], color="tab:blue", alpha=0.18, linewidth=0)
        edge = win.repeat[0] if panel == "left" else win.repeat[1] if panel == "right" else None
        if edge is not None:
            ax.axvline(edge, color="tab:red", linewidth=1.0)
        ax.set_xlim(lo, max(hi, lo + 1))
        ax.set_title(captions[panel], fontsize=9)
        ax.set_xlabel("sample of the read")
        ax.set_ylabel("signal (robust z)")
    return axes


def figure_name(row, fmt):
    return "%s_%s_%s.%s" % (row.target, row.count, row.read_id, fmt)
