"""(De)serialisation of a reference C++ config tree made of ``mettagrid_amd.mettagrid_c`` records — fixture data for the
tests (tests/golden/ref_*.json).  Object identity is part of the data (the converter registers one handler config under
several tag ids and aliases one agent config under several cell names), so every record gets an id and repeats are
references."""
from __future__ import annotations

import enum

from mettagrid_amd import mettagrid_c as C


def dump(obj, seen=None):
    seen = {} if seen is None else seen
    if isinstance(obj, C._Record):
        if id(obj) in seen:
            return {"$ref": seen[id(obj)]}
        seen[id(obj)] = len(seen)
        return {"$class": type(obj).__name__, "$id": seen[id(obj)], "fields": {k: dump(v, seen) for k, v in obj.__dict__.items()}}
    if isinstance(obj, enum.Enum):
        return {"$enum": f"{type(obj).__name__}.{obj.name}"}
    if isinstance(obj, dict):
        return {"$dict": [[dump(k, seen), dump(v, seen)] for k, v in obj.items()]}
    if isinstance(obj, tuple):
        return {"$tuple": [dump(v, seen) for v in obj]}
    if isinstance(obj, list):
        return [dump(v, seen) for v in obj]
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if hasattr(obj, "item"):  # numpy scalars
        return obj.item()
    raise TypeError(f"cannot serialise {type(obj)} in a config tree")


def load(doc, table=None):
    table = {} if table is None else table
    if isinstance(doc, list):
        return [load(v, table) for v in doc]
    if isinstance(doc, dict):
        if "$ref" in doc:
            return table[doc["$ref"]]
        if "$class" in doc:
            obj = getattr(C, doc["$class"]).__new__(getattr(C, doc["$class"]))
            table[doc["$id"]] = obj
            for k, v in doc["fields"].items():
                setattr(obj, k, load(v, table))
            return obj
        if "$enum" in doc:
            cls, member = doc["$enum"].split(".")
            return getattr(getattr(C, cls), member)
        if "$dict" in doc:
            return {load(k, table): load(v, table) for k, v in doc["$dict"]}
        if "$tuple" in doc:
            return tuple(load(v, table) for v in doc["$tuple"])
    return doc
