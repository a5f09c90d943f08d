"""`Match` and the list `search()` returns (matchstrategy.h:35-44; matching.cpp:266-289).

The library hands a frame's matches over as one array of 32-byte records (include/fdcm.h: fdcm_match).  The
reference's Python module returns `list[Match]`; making 27 025 Python objects out of the records costs ~300x what the
GPU frame behind them costs, so the list here is lazy: `MatchList` is a sequence over the record array and a `Match` is
a view of one record, made when it is asked for.  `penalize()` and `sort_matches()` take the records straight from a
`MatchList` (one memcpy, no Python loop) and return a `MatchList`; plain `list[Match]` arguments still work.

Semantics kept from `list[Match]`:
  * `m = matches[i]; m.score = 1.0; m.transform[0, 2] += 3` changes the list's element (a view writes through),
  * slices share their elements with the list they were cut from,
  * any operation that restructures the list (`insert`, `del`, `sort`, `reverse`, item assignment, `append`, ...)
    first turns it into a real Python list of the same `Match` objects and keeps it that way: always correct, just no
    longer the fast path.
"""
from collections.abc import MutableSequence
from itertools import repeat

import numpy as np

from ._capi import MATCH_DTYPE


class _Store:
    """One record array and the column views a Match reads through."""
    __slots__ = ("rec", "idx", "score", "tf")

    def __init__(self, rec):
        self.rec = rec
        self.idx = rec["tmpl_idx"]
        self.score = rec["score"]
        self.tf = rec["transform"].reshape(rec.shape[0], 2, 3)  # a view: the field's rows are contiguous


class Match:
    """matchstrategy.h:35-44: `tmpl_idx`, `score`, `transform` ((2, 3) float32), all read-write."""
    __slots__ = ("_s", "_i")

    def __init__(self, tmpl_idx, score, transform):
        rec = np.empty(1, dtype=MATCH_DTYPE)
        rec["tmpl_idx"][0] = int(tmpl_idx)
        rec["score"][0] = score
        rec["transform"][0] = np.asarray(transform, dtype=np.float32).reshape(6)
        self._s = _Store(rec)
        self._i = 0

    @property
    def tmpl_idx(self):
        return int(self._s.idx[self._i])

    @tmpl_idx.setter
    def tmpl_idx(self, v):
        self._s.idx[self._i] = int(v)

    @property
    def score(self):
        return float(self._s.score[self._i])

    @score.setter
    def score(self, v):
        self._s.score[self._i] = v

    @property
    def transform(self):
        return self._s.tf[self._i]

    @transform.setter
    def transform(self, v):
        self._s.tf[self._i] = np.asarray(v, dtype=np.float32).reshape(2, 3)

    def _record(self):
        return self._s.rec[self._i]

    def __eq__(self, other):
        if not isinstance(other, Match):
            return NotImplemented
        return self._record().tobytes() == other._record().tobytes()

    def __reduce__(self):  # by value: a pickled or deep-copied Match does not drag its list's records along
        return (Match, (self.tmpl_idx, self.score, np.array(self.transform, dtype=np.float32, copy=True)))

    def __hash__(self):  # by value, like __eq__ (the reference's Match hashes by identity: sets and dict keys keep working)
        return hash(self._record().tobytes())

    def __repr__(self):
        return f"<Match tmplIdx={self.tmpl_idx}, score={self.score:g}, transform=\n{self.transform}>"


def _view(store, i):
    m = Match.__new__(Match)
    m._s = store
    m._i = i
    return m


def records_of(matches):
    """The matches as one contiguous record array (MATCH_DTYPE).  A fresh copy: callers may modify it."""
    if isinstance(matches, MatchList) and matches._items is None:
        return np.array(matches._s.rec, dtype=MATCH_DTYPE, copy=True, order="C")
    if isinstance(matches, np.ndarray) and matches.dtype == MATCH_DTYPE:
        return np.array(matches, copy=True, order="C")
    items = matches._items if isinstance(matches, MatchList) else list(matches)
    n = len(items)
    rec = np.empty(n, dtype=MATCH_DTYPE)
    if n:
        # consecutive views of one store (a MatchList turned into a list and passed on): one gather
        first = items[0]
        if isinstance(first, Match) and all(isinstance(m, Match) and m._s is first._s for m in items):
            rec[:] = first._s.rec[np.fromiter((m._i for m in items), dtype=np.int64, count=n)]
            return rec
        rec["tmpl_idx"] = np.fromiter((m.tmpl_idx for m in items), dtype=np.int32, count=n)
        rec["score"] = np.fromiter((m.score for m in items), dtype=np.float32, count=n)
        tf = rec["transform"]
        for i, m in enumerate(items):
            tf[i] = np.asarray(m.transform, dtype=np.float32).reshape(6)
    return rec


class MatchList(MutableSequence):
    """The sequence of `Match` that `search`, `penalize` and `sort_matches` return."""
    __slots__ = ("_s", "_items")

    def __init__(self, records=None):
        if records is None:
            records = np.zeros(0, dtype=MATCH_DTYPE)
        elif not (isinstance(records, np.ndarray) and records.dtype == MATCH_DTYPE):
            records = records_of(records)
        self._s = _Store(records)
        self._items = None  # a real list once the sequence has been restructured

    # ---- the fast path: reads
    def records(self):
        """The structured array behind the list (tmpl_idx, score, transform[6]); a copy once restructured."""
        return self._s.rec if self._items is None else records_of(self._items)

    def __len__(self):
        return self._s.rec.shape[0] if self._items is None else len(self._items)

    def __getitem__(self, i):
        if self._items is not None:
            r = self._items[i]
            return MatchList._of_items(r) if isinstance(i, slice) else r
        if isinstance(i, slice):
            return MatchList(self._s.rec[i])
        n = self._s.rec.shape[0]
        j = i.__index__()
        if j < 0:
            j += n
        if not 0 <= j < n:
            raise IndexError("list index out of range")
        return _view(self._s, j)

    def __iter__(self):
        if self._items is not None:
            return iter(self._items)
        s = self._s
        return map(_view, repeat(s), range(s.rec.shape[0]))

    def __reversed__(self):
        if self._items is not None:
            return reversed(self._items)
        s = self._s
        return map(_view, repeat(s), range(s.rec.shape[0] - 1, -1, -1))

    def __eq__(self, other):
        if isinstance(other, MatchList):
            if self._items is None and other._items is None:
                a, b = self._s.rec, other._s.rec
                return a.shape == b.shape and a.tobytes() == b.tobytes()
            return list(self) == list(other)
        if isinstance(other, list):
            return len(self) == len(other) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    __hash__ = None

    def __add__(self, other):
        if isinstance(other, (MatchList, list)):
            return MatchList(np.concatenate([records_of(self), records_of(other)]))
        return NotImplemented

    def __radd__(self, other):
        if isinstance(other, list):
            return MatchList(np.concatenate([records_of(other), records_of(self)]))
        return NotImplemented

    def copy(self):
        return self[:]

    def __reduce__(self):  # pickle / deepcopy by value: one record array (the column views of the store are rebuilt, not pickled)
        if self._items is None:
            return (MatchList, (np.array(self._s.rec, dtype=MATCH_DTYPE, copy=True, order="C"),))
        return (_matchlist_of_items, (list(self._items),))

    def __copy__(self):  # copy.copy: a new list object over the same elements, like copy.copy(list)
        return self[:]

    def __repr__(self):
        n = len(self)
        if n <= 6:
            return "[" + ", ".join(repr(m) for m in self) + "]"
        return f"<MatchList of {n} matches, best score {float(np.min(self.records()['score'])):g}>"

    # ---- restructuring: become a real list of the same elements
    @classmethod
    def _of_items(cls, items):
        ml = cls()
        ml._items = list(items)
        return ml

    def _materialise(self):
        if self._items is None:
            self._items = list(iter(self))
        return self._items

    def __setitem__(self, i, v):
        self._materialise()[i] = list(v) if isinstance(i, slice) else v

    def __delitem__(self, i):
        del self._materialise()[i]

    def insert(self, i, v):
        self._materialise().insert(i, v)

    def append(self, v):
        self._materialise().append(v)

    def extend(self, vs):
        self._materialise().extend(vs)

    def reverse(self):
        self._materialise().reverse()

    def sort(self, *, key=None, reverse=False):
        self._materialise().sort(key=key, reverse=reverse)



def _matchlist_of_items(items):
    return MatchList._of_items(items)
