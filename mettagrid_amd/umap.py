"""Order emulation of libstdc++ ``std::unordered_map<uint8_t|int, T>`` (identity hash, unique keys).

Why this exists: the reference keeps inventories and several config tables in ``std::unordered_map`` and the
ITERATION ORDER of those maps is observable (observation-token order of inventory items:
/root/reference/cpp/src/mettagrid/core/grid_object.cpp:195-199; initial-inventory insertion order:
/root/reference/cpp/src/mettagrid/core/grid_object_factory.cpp:83-87, objects/agent.cpp:79-84; limit drop order:
objects/inventory.cpp:141-173).  The order is a property of libstdc++'s ``_Hashtable`` (g++ 11, the oracle build):

* nodes live on one singly linked list; all nodes of a bucket are contiguous;
* insert into a non-empty bucket -> node goes to the FRONT of that bucket's run; into an empty bucket -> node goes
  to the global list head (``_M_insert_bucket_begin``);
* erase unlinks the node, nothing else moves;
* growth: ``_Prime_rehash_policy::_M_need_rehash`` (first insert into a default-constructed map -> 13 buckets, then
  29, 59, ...; ``reserve(n)`` uses the small-prime fast table) and ``_M_rehash_aux`` relinks nodes in current
  iteration order with the same two rules.

This is host logic (config compilation); the device never sees a hash table — see DESIGN.md "inventory order".
"""
from __future__ import annotations

import bisect
import math

# libstdc++ __prime_list head (src/shared/hashtable-aux.cc); enough for <= 256 keys.
_PRIMES = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 103,
           109, 113, 127, 137, 139, 149, 157, 167, 179, 193, 199, 211, 227, 241, 257, 277, 293, 313, 337, 359,
           383, 409, 439, 467, 503, 541, 577, 619, 661, 709, 761, 823, 887, 953, 1031]
_FAST_BKT = [2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13]


class UMap:
    """Tracks key order only (values are irrelevant to ordering)."""

    def __init__(self) -> None:
        self.order: list[int] = []  # iteration order, begin() first
        self.nb = 1                 # bucket count
        self.next_resize = 0        # _M_next_resize
        self.values: dict[int, int] = {}

    # -- _Prime_rehash_policy ------------------------------------------------------------------------------
    def _next_bkt(self, n: int) -> int:
        if n < len(_FAST_BKT):
            if n == 0:
                return 1
            self.next_resize = _FAST_BKT[n]
            return _FAST_BKT[n]
        p = _PRIMES[bisect.bisect_left(_PRIMES, n)]
        self.next_resize = p
        return p

    def _need_rehash(self, n_ins: int = 1) -> int | None:
        n_elt = len(self.order)
        if n_elt + n_ins > self.next_resize:
            min_bkts = float(max(n_elt + n_ins, 0 if self.next_resize else 11))
            if min_bkts >= self.nb:
                return self._next_bkt(max(int(math.floor(min_bkts)) + 1, self.nb * 2))
            self.next_resize = self.nb
        return None

    def _rehash(self, nb: int) -> None:
        old = self.order
        self.order = []
        self.nb = nb
        for k in old:
            self._link(k)

    def _link(self, k: int) -> None:
        b = k % self.nb
        for i, other in enumerate(self.order):
            if other % self.nb == b:
                self.order.insert(i, k)  # front of the bucket's run
                return
        self.order.insert(0, k)          # empty bucket: global head

    # -- public ---------------------------------------------------------------------------------------------
    def reserve(self, n: int) -> None:
        """``unordered_map::reserve`` == ``rehash(ceil(n / max_load_factor))`` (pybind11's map caster calls it)."""
        want = max(n, len(self.order) + 1)  # rehash(): max(_M_bkt_for_elements(size + 1), n)
        nb = self._next_bkt(want)
        if nb != self.nb:
            self._rehash(nb)

    def insert(self, k: int, v: int = 0) -> None:
        if k in self.values:
            self.values[k] = v
            return
        nb = self._need_rehash()
        if nb is not None:
            self._rehash(nb)
        self._link(k)
        self.values[k] = v

    def erase(self, k: int) -> None:
        if k in self.values:
            self.order.remove(k)
            del self.values[k]

    def keys(self) -> list[int]:
        return list(self.order)

    def copy(self) -> "UMap":
        """Copy construction keeps bucket count and iteration order (``_M_assign``)."""
        c = UMap()
        c.order = list(self.order)
        c.nb = self.nb
        c.next_resize = self.next_resize
        c.values = dict(self.values)
        return c


def from_pydict(keys: list[int]) -> UMap:
    """Order of a C++ unordered_map produced by pybind11's map caster from a Python dict with these keys
    (``reserve(len)`` then ``emplace`` in dict order; pybind11/stl.h map_caster::load)."""
    m = UMap()
    m.reserve(len(keys))
    for k in keys:
        m.insert(k)
    return m
