#!/usr/bin/env python3
"""Round 6: fix the A/B switches of mjx_coop.h at the values every shipped build uses and delete the branches that are never compiled
(the measurements stay in docs/rejected_experiments.md and docs/mujoco_design.md, the code in the git history).

    python scripts/r06/prune_macros.py gymnasium_amd/csrc/mjx_coop.h

A tiny partial preprocessor: `#if` / `#elif` conditions that mention ONLY the fixed macros (and integer literals, !, ==, !=, &&, ||, parentheses)
are evaluated and their dead branches removed; every other conditional is left alone; the `#ifndef X / #define X v / #endif` blocks of the fixed
macros go; remaining uses of a fixed macro in ordinary code are replaced by its value."""
import re
import sys

FIXED = {"MJX_CHOL_LDS_FOR_16": 0, "MJX_KIN_LOCAL_JOINTS": 1, "MJX_FLAT_JOINTS": 1, "MJX_RK4_INLINE": 0, "MJX_PGS_MORE_BLOCKS": 1, "MJX_CRB_BLEND_ALL": 1,
         "MJX_CRB_BRANCHFREE": 1, "MJX_COLLIDE_TABLES": 1, "MJX_CHOL_PIPELINED": 1, "MJX_VEL_PREFIX": 1, "MJX_KIN_PREFIX": 1, "MJX_CHOL_MFMA": 0,
         "MJX_PGS_PIPELINE": 0, "MJX_PGS_EDGE_CHAIN": 0, "MJX_PGS_QS_BY_INVERSE": 0, "MJX_GROUP_SUM_SHFL": 0, "MJX_CHOL_LDS_FOR_32": 1,
         "MJX_SOLVE_BCAST_FOR_32": 1}


def evaluate(cond):
    """Value of a preprocessor condition if it only involves fixed macros, else None."""
    cond = re.sub(r"//.*$", "", cond).strip()
    names = set(re.findall(r"[A-Za-z_]\w*", cond))
    if not names or not names <= set(FIXED):
        return None
    expr = cond
    for n in names:
        expr = re.sub(rf"\b{n}\b", str(FIXED[n]), expr)
    expr = expr.replace("&&", " and ").replace("||", " or ")
    expr = re.sub(r"!(?!=)", " not ", expr)
    if not re.fullmatch(r"[\d\s()=!<>andortn]+", expr):
        return None
    return bool(eval(expr))


def prune(lines):
    out, i = [], 0
    # stack entries: dict(kind="eval"|"keep", taken=bool (a branch already emitted), emitting=bool)
    stack = []

    def emitting():
        return all(f["emitting"] for f in stack)

    while i < len(lines):
        ln = lines[i]
        m = re.match(r"\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", ln)
        if not m:
            if emitting():
                out.append(ln)
            i += 1
            continue
        d, rest = m.group(1), m.group(2)
        if d == "ifndef" and rest.strip().split()[0] in FIXED:  # the definition block of a fixed macro: drop it whole
            j = i + 1
            while not re.match(r"\s*#\s*endif", lines[j]):
                j += 1
            i = j + 1
            continue
        if d in ("if", "ifdef", "ifndef"):
            v = evaluate(rest) if d == "if" else None
            if v is None:
                stack.append({"kind": "keep", "emitting": True})
                if emitting():
                    out.append(ln)
            else:
                stack.append({"kind": "eval", "taken": v, "emitting": v})
        elif d == "elif":
            f = stack[-1]
            if f["kind"] == "keep":
                if emitting():
                    out.append(ln)
            else:
                v = evaluate(rest)
                assert v is not None, f"mixed #elif at line {i + 1}: {ln}"
                f["emitting"] = (not f["taken"]) and v
                f["taken"] = f["taken"] or v
        elif d == "else":
            f = stack[-1]
            if f["kind"] == "keep":
                if emitting():
                    out.append(ln)
            else:
                f["emitting"] = not f["taken"]
                f["taken"] = True
        else:  # endif
            f = stack.pop()
            if f["kind"] == "keep" and emitting():
                out.append(ln)
        i += 1
    assert not stack
    return out


def main():
    path = sys.argv[1]
    src = open(path).read().split("\n")
    res = prune(src)
    text = "\n".join(res)
    for n, v in FIXED.items():  # uses outside the preprocessor (constexpr flags, `if (MJX_X)`)
        text = re.sub(rf"\b{n}\b", str(v), text)
    open(path, "w").write(text)
    print(f"{path}: {len(src)} -> {len(res)} lines")


if __name__ == "__main__":
    main()
