"""Loads the constants of ``include/mgx_program.h`` so Python and C/HIP share one definition of the program format."""
from __future__ import annotations

import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "mgx_program.h")


def _strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def load_constants(path: str = HEADER) -> dict:
    text = _strip_comments(open(path).read())
    env: dict = {}
    for m in re.finditer(r"#define\s+(MGX_\w+)\s+\(?(-?(?:0x)?[0-9a-fA-F]+)\)?", text):
        env[m.group(1)] = int(m.group(2), 0)
    for m in re.finditer(r"enum\s*\{(.*?)\}", text, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, expr = [x.strip() for x in item.split("=", 1)]
                nxt = int(eval(expr, {"__builtins__": {}}, env))  # noqa: S307 - header constants only
            else:
                name = item
            env[name] = nxt
            nxt += 1
    return env


C = load_constants()


class _NS:
    def __init__(self, d: dict) -> None:
        for k, v in d.items():
            setattr(self, k[4:] if k.startswith("MGX_") else k, v)


K = _NS(C)  # K.H_MAGIC, K.SEC_CLASSES, K.FOP_VIBE ...
