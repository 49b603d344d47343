"""Per-forward memo tables (values every block of one forward recomputes identically: the padded text context, the camera constants of
the view logits, the Plucker features of a level) and the one rule that keeps them honest around hipGraph capture.

An entry is keyed on the (tensor object, version) it was derived from.  That is not enough while a stream is being captured: an entry
made by an EAGER run with the same key would be served to the capture, the kernels that produce it would not be part of the graph, and a
later replay -- after the source buffer was rewritten in place (bench.py's Sampler.retarget points a captured sampler at the next pose
this way) -- would read the stale eager value.  Likewise an entry made inside capture A must not feed capture B (graph B would depend on
graph A having been replayed first).  So every entry carries the capture EPOCH it was made in -- the epoch advances whenever the
capturing state of the current stream flips -- and is served only to the same epoch, or eager-to-eager."""
from __future__ import annotations

import torch

_state = {"capturing": False, "epoch": 0}


def _now():
    cap = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    if cap != _state["capturing"]:
        _state["capturing"] = cap
        _state["epoch"] += 1
    return cap, _state["epoch"]


def new_epoch() -> None:
    """Advance the capture epoch explicitly.  The polled flip in `_now` is only seen when some `Memo.get` happens to run between two
    captures; a capture helper calls this right before it begins capturing and right after it ends, so that two captures with no eager
    lookup between them (two GraphedTrainStep(warmup=0) in a row) can never share an epoch."""
    _state["epoch"] += 1


class Memo:
    """The last `keep` (source tensor, version) -> value entries.  `get(src, make)` returns the memoised value or makes it."""

    def __init__(self, keep: int = 4):
        self.keep, self.entries = keep, []

    def get(self, src: torch.Tensor, make, extra=None):
        cap, epoch = _now()
        for ent in self.entries:
            if ent[0] is src and ent[1] == src._version and ent[2] == extra and (ent[4] == epoch or not (cap or ent[3])):
                return ent[5]
        val = make()
        self.entries.insert(0, (src, src._version, extra, cap, epoch, val))
        del self.entries[self.keep:]
        return val

    def clear(self):
        self.entries.clear()
