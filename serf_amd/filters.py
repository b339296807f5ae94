"""QueryParam.filters on the host side (reference: serf-core/src/types/filter.rs, should_process_query query.rs:439-521).

A filter is either ``Filter.id([...node ids])`` — only those nodes process the query — or ``Filter.tag(name, expr)`` —
only nodes whose tag `name` exists and whose value matches the regular expression.  A node processes a query iff EVERY
filter matches.

Regular expressions over tag strings are host work: the simulated nodes carry a TAG CLASS (0..31) instead of a tag
map, class 0 being "no tags".  `TagTable` keeps the distinct tag maps in use, `compile` evaluates each tag filter
once per class and reduces a filter list to what the ABI takes (include/serf_sim.h, sim_query_filtered): one id list
(the intersection of the Filter::Id lists) and one mask of tag classes (the AND of the per-filter masks).
"""
from __future__ import annotations

import re
from dataclasses import dataclass

from . import _ffi


@dataclass(frozen=True)
class Filter:
    kind: str           # "id" | "tag"
    ids: tuple = ()
    tag: str = ""
    expr: str = ""

    @staticmethod
    def id(ids):
        return Filter("id", ids=tuple(int(i) for i in ids))

    @staticmethod
    def tag(name, expr):
        re.compile(expr)  # "invalid regex" surfaces where the reference reports it: at the caller
        return Filter("tag", tag=name, expr=expr)


class TagTable:
    """The distinct tag maps of a simulated cluster and the class each one got."""

    def __init__(self):
        self.classes = [{}]  # class 0: no tags

    def class_of(self, tags: dict) -> int:
        tags = dict(tags)
        if tags in self.classes:
            return self.classes.index(tags)
        if len(self.classes) == _ffi.TAG_CLASSES:
            raise ValueError(f"more than {_ffi.TAG_CLASSES - 1} distinct tag sets (model bound)")
        self.classes.append(tags)
        return len(self.classes) - 1

    def mask(self, flt: Filter) -> int:
        """Classes a Filter::Tag matches (query.rs:463-481 / 497-515): tags non-empty, the tag present, the expression
        matching its value anywhere (Regex::is_match is unanchored, like re.search)."""
        rx = re.compile(flt.expr)
        m = 0
        for c, tags in enumerate(self.classes):
            if tags and flt.tag in tags and rx.search(tags[flt.tag]):
                m |= 1 << c
        return m

    def compile(self, filters):
        """-> (ids or None, tag_mask) for Sim.query.  An empty id intersection cannot be told from "no id filter" in
        the ABI (n_ids = 0 means none), so it comes back as tag_mask = 0: nobody processes the query."""
        ids, mask = None, _ffi.NO_TAG_FILTER
        for f in filters:
            if f.kind == "id":
                ids = list(f.ids) if ids is None else [i for i in ids if i in f.ids]
            else:
                mask &= self.mask(f)
        if ids is not None:
            ids = list(dict.fromkeys(ids))
            if not ids:
                return None, 0
            if len(ids) > _ffi.QF_IDS:
                raise ValueError(f"Filter::Id with more than {_ffi.QF_IDS} nodes: give them a tag and filter on it (model bound)")
        return ids, mask
