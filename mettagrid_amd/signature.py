"""Generalised deterministic episode signature.

The reference's parity script (/root/reference/scripts/deterministic_episode_signature.py:19-47,97-110) hashes a
JSON payload of objects + stats + rewards for ONE fixed scenario.  This module restates the payload function so it
can be applied to any engine state: the reference (through its ``grid_objects()`` / ``get_episode_stats()``), the CPU
oracle and the HIP engine (through their raw object/stat dumps).  Same schema, same rounding (8 decimals), same
``json.dumps(sort_keys=True, separators=(",", ":"))`` + SHA-256.
"""
from __future__ import annotations

import hashlib
import json

import numpy as np

_MAXR = 13


def _round(v) -> float:
    return round(float(v), 8)


def stats_dicts(prog, gv, gt, av, at, extra=None) -> dict:
    """Raw stat arrays + touched flags -> {"game": {name: value}, "agent": [{...}]} (keys exist once touched:
    cpp/include/mettagrid/systems/stats_tracker.hpp:57-67,109-115)."""
    game = {prog.game_stat_names[i]: float(gv[i]) for i in range(len(gv)) if gt[i]}
    agents = [{prog.agent_stat_names[i]: float(av[a, i]) for i in range(av.shape[1]) if at[a, i]}
              for a in range(av.shape[0])]
    for a, more in enumerate(extra or []):   # invalid action indices without a stat column: {k: count} per agent
        for k, n in more.items():
            agents[a][f"action.invalid_index.{k}"] = float(n)
    return {"game": game, "agent": agents}


def objects_from_raw(prog, raw: np.ndarray, current_stat_reward=None) -> dict:
    """Raw object records (see mgx.h mgx_get_objects) -> reference ``grid_objects()``-shaped dict
    (cpp/bindings/mettagrid_py.cpp:28-139; only the fields the signature and tests use)."""
    from .fmt import K
    words = prog.words
    coff = int(words[K.H_SECTION_BASE + 2 * K.SEC_CLASSES])
    out = {}
    for rec in raw:
        oid, cls, r, c, vibe, alive, agent_id, norder = (int(x) for x in rec[:8])
        if not alive:
            continue
        C = words[coff + cls * K.C_WORDS: coff + (cls + 1) * K.C_WORDS]
        tw = rec[8 + 2 * _MAXR: 8 + 2 * _MAXR + 8]
        tags = [t for t in range(256) if (int(tw[t >> 5]) >> (t & 31)) & 1]
        amounts = rec[8 + _MAXR: 8 + 2 * _MAXR]
        d = {"id": oid, "type_name": prog.type_names[int(C[K.C_TYPE_ID])], "r": r, "c": c, "location": (c, r),
             "tag_ids": tags, "vibe": vibe,
             "inventory": {i: int(amounts[i]) for i in range(_MAXR) if amounts[i] > 0},
             "inventory_order": [int(x) for x in rec[8:8 + norder]]}
        if agent_id >= 0:
            d["agent_id"] = agent_id
            d["group_id"] = int(C[K.C_GROUP])
            d["current_stat_reward"] = float(current_stat_reward[agent_id]) if current_stat_reward is not None else 0.0
        out[oid] = d
    return out


def payload(objects: dict, stats: dict, action_success, episode_rewards, steps: int, seed: int) -> dict:
    objs = []
    for obj_id, obj in sorted(objects.items()):
        entry = {"id": int(obj_id), "type_name": obj["type_name"], "location": [int(obj["r"]), int(obj["c"])],
                 "tag_ids": [int(t) for t in obj.get("tag_ids", [])],
                 "inventory_items": [(int(k), int(v)) for k, v in sorted(obj.get("inventory", {}).items())]}
        if "agent_id" in obj:
            entry["agent_id"] = int(obj["agent_id"])
            entry["group_id"] = int(obj["group_id"])
            entry["vibe"] = int(obj["vibe"])
            entry["current_stat_reward"] = _round(obj["current_stat_reward"])
        objs.append(entry)
    return {
        "seed": int(seed), "steps": int(steps), "action_success": [bool(x) for x in action_success],
        "episode_reward": [_round(v) for v in episode_rewards], "objects": objs,
        "stats": {"game": [(n, _round(v)) for n, v in sorted(stats["game"].items())],
                  "agent": [[(n, _round(v)) for n, v in sorted(a.items())] for a in stats["agent"]]},
    }


def signature(p: dict) -> str:
    return hashlib.sha256(json.dumps(p, sort_keys=True, separators=(",", ":")).encode("utf-8")).hexdigest()


# ---- batched state digest (include/mgx.h mgx_state_digests) ------------------------------------------------------------
_FNV_BASIS, _FNV_PRIME, _M64 = 14695981039346656037, 1099511628211, (1 << 64) - 1


def _fnv(h: int, w: int) -> int:
    return ((h ^ (int(w) & 0xFFFFFFFF)) * _FNV_PRIME) & _M64


def state_digest(raw_objects: np.ndarray, raw_stats, episode_rewards, action_success, current_stat_reward, step: int,
                 invalid_extra=None) -> int:
    """The digest mgx_state_digests computes for one env, from the raw dumps every engine offers (the HIP engine through
    mgx_get_objects / mgx_get_stats / ..., the CPU oracle through its mgxo_* twins)."""
    gv, gt, av, at = raw_stats
    h = _fnv(_fnv(_FNV_BASIS, step), len(raw_objects))
    for rec in raw_objects:
        hs = _FNV_BASIS
        for w in rec:
            hs = _fnv(hs, int(w))
        h = _fnv(_fnv(h, hs & 0xFFFFFFFF), hs >> 32)
    bits = lambda x: int(np.float32(x).view(np.uint32))  # noqa: E731
    for i in range(len(gv)):
        h = _fnv(_fnv(h, bits(gv[i])), int(bool(gt[i]) or gv[i] != 0))
    for a in range(av.shape[0]):
        for i in range(av.shape[1]):
            h = _fnv(_fnv(h, bits(av[a, i])), int(bool(at[a, i]) or av[a, i] != 0))
        h = _fnv(h, bits(episode_rewards[a]))
        h = _fnv(h, int(bool(action_success[a])))
        h = _fnv(h, bits(current_stat_reward[a]))
        for k, n in (invalid_extra[a] if invalid_extra else {}).items():   # invalid_index_extra(): first-seen order
            h = _fnv(_fnv(h, int(k) & 0xFFFFFFFF), bits(n))
    return h


def state_digest_many(dumps: list) -> np.ndarray:
    """``state_digest`` of many envs at once (uint64 [N]): the FNV chains run side by side in numpy, one array step per
    hashed word.  ``dumps``: per env ``(raw_objects, raw_stats, episode_rewards, action_success, current_stat_reward, step)``
    with an optional seventh entry ``invalid_extra``; envs that have such extras take the scalar function."""
    N = len(dumps)
    out = np.zeros(N, np.uint64)
    plain = [i for i, d in enumerate(dumps) if not (len(d) > 6 and d[6] and any(d[6]))]
    for i in set(range(N)) - set(plain):
        out[i] = state_digest(*dumps[i])
    if not plain:
        return out
    D = [dumps[i] for i in plain]
    n = len(D)
    prime, m32 = np.uint64(_FNV_PRIME), np.uint64(0xFFFFFFFF)

    def fnv(h, w):
        return (h ^ (w.astype(np.uint64) & m32)) * prime

    with np.errstate(over="ignore"):
        nobj = np.array([len(d[0]) for d in D], np.int64)
        words = D[0][0].shape[1] if nobj.max() > 0 else 0
        recs = np.zeros((n, int(nobj.max()), words), np.int64)
        for i, d in enumerate(D):
            recs[i, :nobj[i]] = d[0]
        hs = np.full(recs.shape[:2], _FNV_BASIS, np.uint64)
        for w in range(words):
            hs = fnv(hs, recs[:, :, w].astype(np.uint64))
        h = fnv(fnv(np.full(n, _FNV_BASIS, np.uint64), np.array([d[5] for d in D], np.uint64)), nobj.astype(np.uint64))
        for k in range(recs.shape[1]):
            live = k < nobj
            h = np.where(live, fnv(fnv(h, hs[:, k] & m32), hs[:, k] >> np.uint64(32)), h)
        f32bits = lambda x: np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)  # noqa: E731
        gv = np.stack([f32bits(d[1][0]) for d in D])
        gt = np.stack([(np.asarray(d[1][1]) != 0) | (np.asarray(d[1][0]) != 0) for d in D])
        for i in range(gv.shape[1]):
            h = fnv(fnv(h, gv[:, i]), gt[:, i])
        av = np.stack([f32bits(d[1][2]) for d in D])
        at = np.stack([(np.asarray(d[1][3]) != 0) | (np.asarray(d[1][2]) != 0) for d in D])
        ep = np.stack([f32bits(d[2]) for d in D])
        ok = np.stack([np.asarray(d[3]).astype(bool) for d in D])
        cur = np.stack([f32bits(d[4]) for d in D])
        for a in range(av.shape[1]):
            for i in range(av.shape[2]):
                h = fnv(fnv(h, av[:, a, i]), at[:, a, i])
            h = fnv(fnv(fnv(h, ep[:, a]), ok[:, a]), cur[:, a])
    out[plain] = h
    return out
