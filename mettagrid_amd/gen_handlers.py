"""build()-time specialisation of the handler interpreter: straight-line HIP code for the handlers of the benchmark presets.

The world kernels execute a program's handlers (filters -> mutations, multi-handlers, UseTarget) on a small interpreter
(csrc/mgx_world.h ``run_handler``): an explicit frame stack, a stage machine, a record fetch and a switch per filter atom
and mutation.  For a FIXED program all of that is known at build time.  ``render()`` walks the compiled preset programs
(BASELINE.json configs[2] and configs[3]) and writes ``csrc/mgx_handlers_gen.h``: one function per handler and nesting
level, filter chains as branches, mutation operands as constants (the shared ``atom_v`` / ``mutate_v`` switches fold), calls
where the interpreter pushed frames.  Semantics are ``run_handler``'s, statement for statement (handler/handler.cpp:76-103,
multi_handler.cpp:8-21, use_target_mutation.hpp:17-29).

The engine uses the generated code only for a program whose handler tables hash to the fingerprint recorded here
(``mgx_create``: MgxDev::gen_prog); every other program — and a preset edited without a rebuild — runs the interpreter.
"""
from __future__ import annotations

import os

import numpy as np

from . import presets
from .compiler import compile_spec
from .fmt import K

_HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(_HERE, "csrc", "mgx_handlers_gen.h")
MAX_FRAMES = 4   # run_handler keeps four frames


class _Prog:
    def __init__(self, prog):
        self.w = prog.words
        w = self.w
        self.sec = lambda s: int(w[K.H_SECTION_BASE + 2 * s])
        self.cnt = lambda s: int(w[K.H_SECTION_BASE + 2 * s + 1])
        self.handlers = [self._rec(K.SEC_HANDLERS, i, K.HD_WORDS) for i in range(self.cnt(K.SEC_HANDLERS))]
        self.children = [int(x) for x in w[self.sec(K.SEC_CHILDREN): self.sec(K.SEC_CHILDREN) + self.cnt(K.SEC_CHILDREN)]]
        self.atoms = [self._rec(K.SEC_ATOMS, i, K.AT_WORDS) for i in range(self.cnt(K.SEC_ATOMS))]
        self.muts = [self._rec(K.SEC_MUTS, i, K.MU_WORDS) for i in range(self.cnt(K.SEC_MUTS))]
        self.move = [self._rec(K.SEC_MOVE_HANDLERS, i, K.MH_WORDS) for i in range(self.cnt(K.SEC_MOVE_HANDLERS))]
        nc = int(w[K.H_NUM_CLASSES])
        cls = [self._rec(K.SEC_CLASSES, c, K.C_WORDS) for c in range(nc)]
        self.class_hooks = [(c[K.C_ON_USE], c[K.C_ON_AFTER_USE], c[K.C_ON_TICK]) for c in cls]
        self.game_on_tick = int(w[K.H_GAME_ON_TICK])

    def _rec(self, section, i, words):
        o = self.sec(section) + i * words
        return [int(x) for x in self.w[o: o + words]]

    def fingerprint(self) -> int:
        """FNV-1a over 32-bit words; the same walk as mgx_engine.hip mgx_handler_fingerprint."""
        h = 0xCBF29CE484222325
        def mix(v):
            nonlocal h
            h = ((h ^ (v & 0xFFFFFFFF)) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        for recs in (self.handlers, [[c] for c in self.children], self.atoms, self.muts, self.move):
            mix(len(recs))
            for r in recs:
                for v in r:
                    mix(v)
        mix(len(self.class_hooks))
        for hook in self.class_hooks:
            for v in hook:
                mix(v)
        mix(self.game_on_tick)
        return h


def _closure(p: _Prog, roots) -> list:
    seen, todo = [], [h for h in roots if h >= 0]
    while todo:
        h = todo.pop()
        if h in seen:
            continue
        seen.append(h)
        hd = p.handlers[h]
        if hd[K.HD_KIND] != K.HK_LEAF:
            todo += p.children[hd[K.HD_CHILD_START]: hd[K.HD_CHILD_START] + hd[K.HD_CHILD_COUNT]]
    return sorted(seen)


def _uses_target(p: _Prog, hs) -> bool:
    return any(p.muts[p.handlers[h][K.HD_MUT_START] + i][K.MU_OP] == K.MOP_USE_TARGET
               for h in hs if p.handlers[h][K.HD_KIND] == K.HK_LEAF for i in range(p.handlers[h][K.HD_MUT_COUNT]))


def _levels(p: _Prog):
    """Handlers by nesting level of UseTarget: level 0 = what apply_top is called with, level k + 1 = on_use / on_after_use
    handlers reached from level k.  None when a path needs more frames than run_handler has."""
    roots = [m[K.MH_HANDLER] for m in p.move] + [t for (_, _, t) in p.class_hooks] + [p.game_on_tick]
    use_roots = [u for (u, _, _) in p.class_hooks] + [a for (_, a, _) in p.class_hooks]
    levels = [_closure(p, roots)]
    while _uses_target(p, levels[-1]):
        if len(levels) >= MAX_FRAMES:
            return None
        levels.append(_closure(p, use_roots))

    def frames(h, lvl):   # frames in use while handler h of level lvl runs
        hd = p.handlers[h]
        if hd[K.HD_KIND] != K.HK_LEAF:
            kids = p.children[hd[K.HD_CHILD_START]: hd[K.HD_CHILD_START] + hd[K.HD_CHILD_COUNT]]
            return 1 + max([frames(k, lvl) for k in kids] + [0])
        deeper = 0
        if _uses_target(p, [h]) and lvl + 1 < len(levels):
            deeper = max([frames(k, lvl + 1) for k in levels[lvl + 1]] + [0])
        return 1 + deeper
    if max([frames(h, 0) for h in levels[0]] + [0]) > MAX_FRAMES:
        return None
    return levels


def _filter_fn(p: _Prog, pc: int) -> list:
    """Short-circuit filter code from atom `pc` on (handler/handler.cpp:95-103) as labelled branches."""
    out = [f"  static __device__ __forceinline__ bool flt_{pc}(const Env& e, const MgxCtx& c) {{"]
    seen, todo = [], [pc]
    while todo:
        a = todo.pop()
        if a < 0 or a in seen:
            continue
        seen.append(a)
        todo += [p.atoms[a][K.AT_ON_TRUE], p.atoms[a][K.AT_ON_FALSE]]
    lab = lambda t: "PASS" if t == K.PC_PASS else "FAIL" if t < 0 else f"A{t}"   # noqa: E731
    out.append(f"    goto A{pc};")
    for a in sorted(seen):
        r = p.atoms[a]
        out.append(f"  A{a}: if (e.template atom_v<0>({r[K.AT_OP]}, {r[K.AT_A0]}, {r[K.AT_A1]}, {r[K.AT_A2]}, c, 0)) goto {lab(r[K.AT_ON_TRUE])}; "
                   f"else goto {lab(r[K.AT_ON_FALSE])};")
    out += ["  PASS: return true;", "  FAIL: return false;", "  }"]
    return out


def _handler_fn(p: _Prog, h: int, lvl: int, levels) -> list:
    hd = p.handlers[h]
    sig = f"  static __device__ __forceinline__ bool h{h}_L{lvl}(const Env& e, MgxCtx& c, uint32_t& failed, int cs) {{"
    out = [sig]
    if hd[K.HD_KIND] != K.HK_LEAF:   # MultiHandler::try_apply (multi_handler.cpp:8-21)
        kids = p.children[hd[K.HD_CHILD_START]: hd[K.HD_CHILD_START] + hd[K.HD_CHILD_COUNT]]
        first = hd[K.HD_KIND] == K.HK_FIRST_MATCH
        out.append("    bool any = false;")
        for k in kids:
            out.append(f"    if (h{k}_L{lvl}(e, c, failed, cs)) {{ {'return true;' if first else 'any = true;'} }}")
        out += ["    return any;", "  }"]
        return out
    pc = hd[K.HD_FILTER_PC]
    if pc >= 0:
        out.append(f"    if (!flt_{pc}(e, c)) return false;")
    elif pc != K.PC_PASS:
        out.append("    return false;")
    out.append("    failed &= ~(1u << cs);")
    for i in range(hd[K.HD_MUT_COUNT]):
        mi = hd[K.HD_MUT_START] + i
        m = p.muts[mi]
        op = m[K.MU_OP]
        if op == K.MOP_USE_TARGET:   # use_target_mutation.hpp:17-29, core/grid_object.cpp:72-80
            out.append("    {")
            out.append("      int h2 = -1;")
            out.append("      if (c.target >= 0 && c.actor >= 0 && (c.actor == e.cur_slot || e.agent_of(c.actor) >= 0)) h2 = e.cls_of(c.target)[MGX_C_ON_USE];")
            if lvl + 1 < len(levels):
                out.append("      if (h2 < 0) failed |= 1u << cs;")
                out.append(f"      else if (!run_L{lvl + 1}(e, h2, c, failed, cs + 1)) failed |= 1u << cs;")
                out.append("      else {")
                out.append("        const int after = e.cls_of(c.actor)[MGX_C_ON_AFTER_USE];")
                out.append(f"        if (after >= 0) (void)run_L{lvl + 1}(e, after, c, failed, cs);")
                out.append("      }")
            else:
                out.append("      (void)h2;")
                out.append("      failed |= 1u << cs;   // no class of this program can be used")
            out.append("    }")
        elif op < K.MOP_GAME_VALUE:
            out.append(f"    c.mutation_failed = false; e.mutate_v({op}, {m[K.MU_A0]}, {m[K.MU_A1]}, {m[K.MU_A2]}, {m[K.MU_A3]}, {m[K.MU_A4]}, c); "
                       "if (c.mutation_failed) failed |= 1u << cs;")
        else:
            out.append(f"    c.mutation_failed = false; e.mutate(e.prog() + e.d.sec[MGX_SEC_MUTS] + {mi} * MGX_MU_WORDS, c); "
                       "if (c.mutation_failed) failed |= 1u << cs;")
        out.append("    if ((failed >> cs) & 1u) return false;")
    out += ["    return true;", "  }"]
    return out


def render_one(name: str, prog) -> tuple:
    p = _Prog(prog)
    levels = _levels(p)
    lines = [f"template <class Env> struct MgxGen{name} {{"]
    if levels is None:
        lines += ["  static __device__ __forceinline__ bool top(const Env& e, int h, MgxCtx& c) { return e.run_handler(h, c); }", "};"]
        return lines, 0
    pcs = sorted({p.handlers[h][K.HD_FILTER_PC] for lv in levels for h in lv if p.handlers[h][K.HD_KIND] == K.HK_LEAF and p.handlers[h][K.HD_FILTER_PC] >= 0})
    for pc in pcs:
        lines += _filter_fn(p, pc)
    for lvl in reversed(range(len(levels))):
        for h in levels[lvl]:
            lines += _handler_fn(p, h, lvl, levels)
        lines.append(f"  static __device__ __forceinline__ bool run_L{lvl}(const Env& e, int h, MgxCtx& c, uint32_t& failed, int cs) {{")
        lines.append("    switch (h) {")
        for h in levels[lvl]:
            lines.append(f"      case {h}: return h{h}_L{lvl}(e, c, failed, cs);")
        lines += ["      default: e.flag(4u); return false;", "    }", "  }"]
    lines += ["  static __device__ __forceinline__ bool top(const Env& e, int h, MgxCtx& c) {",
              "    uint32_t failed = 0;",
              "    const bool rv = run_L0(e, h, c, failed, 0);",
              "    c.mutation_failed = (failed & 1u) != 0;",
              "    return rv;", "  }", "};"]
    return lines, p.fingerprint()


def render() -> str:
    r3 = compile_spec(presets.rung3_spec(), 32, 32, max_objects=192)
    r4 = compile_spec(presets.rung4_spec(), 64, 64, max_objects=presets.RUNG4_MAX_OBJECTS)
    lines = ["// GENERATED by mettagrid_amd/gen_handlers.py at build() — straight-line code for the handlers of the benchmark presets",
             "// (BASELINE.json configs[2] = R3, configs[3] = R4); included by mgx_world.h inside the unit's namespace.",
             "// Semantics: MgxEnvT::run_handler.  Used only for programs whose handler tables hash to MGX_GEN_*_FP.",
             "#ifndef MGX_HANDLERS_GEN_H_", "#define MGX_HANDLERS_GEN_H_"]
    fps = {}
    for name, prog in (("R3", r3), ("R4", r4)):
        body, fp = render_one(name, prog)
        lines += body
        fps[name] = fp
    lines.append("#endif  // MGX_HANDLERS_GEN_H_")
    lines.append("")
    fp_lines = ["// GENERATED by mettagrid_amd/gen_handlers.py — fingerprints of the handler tables mgx_handlers_gen.h was generated from",
                "#ifndef MGX_HANDLERS_FP_H_", "#define MGX_HANDLERS_FP_H_"]
    for name, fp in fps.items():
        fp_lines.append(f"#define MGX_GEN_{name}_FP 0x{fp:016X}ull")
    fp_lines += ["#endif", ""]
    return "\n".join(lines), "\n".join(fp_lines)


def main() -> None:
    text, fps = render()
    for path, body in ((OUT, text), (os.path.join(_HERE, "csrc", "mgx_handlers_fp.h"), fps)):
        if not os.path.exists(path) or open(path).read() != body:
            open(path, "w").write(body)


if __name__ == "__main__":
    main()
