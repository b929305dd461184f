"""TEST INFRASTRUCTURE (oracle/simt): two rewrites of a reference source, nothing else is touched.
1. The CUDA kernel-launch syntax, so that g++ can parse it:
    kernel<T...> <<<grid, block, shmem, stream>>>(args);   ->   simt::launch(grid, block, [&]() { kernel<T...>(args); });
2. With --converge (every one-lane section of the file: the cudapoa sources) or --converge-lines=<file>:<lines> (the named sections:
   hirschberg_myers_gpu.cu, whose other one-lane sections sit inside divergent code, where a full-warp meeting point would be
   wrong). Reconvergence points around one-lane sections:
    if (lane_idx == 0) {...} [else ...]   ->   { simt::converge(); if (lane_idx == 0) {...} [else ...] simt::converge(); }
   A warp of the GPU executes converged code in lockstep: every lane has finished the statements before a divergent `if` when the
   first lane enters it, and the other lanes wait at its end. The reference relies on that without a __syncwarp() (e.g.
   cudapoa_nw_banded.cuh: all lanes' initialize_band() stores land before lane 0's set_score() of the same cell). The emulator
   runs a lane until its next rendezvous, so the two implicit reconvergence points are made explicit.
usage: python cuda_to_simt.py [--converge] <in> <out>. The output is a build intermediate under oracle/_ref/ that
oracle/Makefile.ref deletes after compiling (reference sources are never copied into the repository)."""
import os
import re
import sys


def matching(text, at, open_ch, close_ch):
    depth = 0
    for i in range(at, len(text)):
        if text[i] == open_ch:
            depth += 1
        elif text[i] == close_ch:
            depth -= 1
            if depth == 0:
                return i
    raise ValueError("unbalanced %s at %d" % (open_ch, at))


def callee_start(text, end):
    """Start of the expression `name<template args>` that ends right before `end` (whitespace skipped)."""
    i = end
    while i > 0 and text[i - 1].isspace():
        i -= 1
    if text[i - 1] == ">":  # template argument list: walk back to its '<'
        depth = 0
        while i > 0:
            i -= 1
            if text[i] == ">":
                depth += 1
            elif text[i] == "<":
                depth -= 1
                if depth == 0:
                    break
    while i > 0 and (text[i - 1].isalnum() or text[i - 1] in "_:"):
        i -= 1
    return i


def rewrite(text):
    out, pos = [], 0
    while True:
        m = text.find("<<<", pos)
        if m < 0:
            break
        close = text.index(">>>", m)
        cfg = text[m + 3:close]
        paren = text.index("(", close)
        assert text[close + 3:paren].strip() == "", text[close:paren + 1]
        end = matching(text, paren, "(", ")")
        start = callee_start(text, m)
        callee = text[start:m].rstrip()
        parts = [p.strip() for p in re.split(r",(?![^()]*\))", cfg)]
        out.append(text[pos:start])
        out.append("simt::launch(%s, %s, [&]() { %s%s; })" % (parts[0], parts[1], callee, text[paren:end + 1]))
        pos = end + 1
    out.append(text[pos:])
    return "".join(out)


ONE_LANE = re.compile(r"\bif\s*\(\s*(lane_idx|threadIdx\.x)\s*==\s*0\s*\)")


def skip_space_and_comments(text, i):
    while i < len(text):
        if text[i].isspace():
            i += 1
        elif text.startswith("//", i):
            i = text.index("\n", i)
        elif text.startswith("/*", i):
            i = text.index("*/", i) + 2
        else:
            break
    return i


def statement_end(text, i):
    """End (exclusive) of the statement that starts at text[i]: a block, an if / else chain, or a simple statement."""
    i = skip_space_and_comments(text, i)
    if text[i] == "{":
        return matching(text, i, "{", "}") + 1
    m = re.match(r"if\s*\(", text[i:])
    if m:
        close = matching(text, i + m.end() - 1, "(", ")")
        end = statement_end(text, close + 1)
        j = skip_space_and_comments(text, end)
        if text.startswith("else", j) and not (text[j + 4].isalnum() or text[j + 4] == "_"):
            return statement_end(text, j + 4)
        return end
    return text.index(";", i) + 1


def add_convergence(text, only_lines=None):
    out, pos = [], 0
    for m in ONE_LANE.finditer(text):
        if m.start() < pos:
            continue
        if only_lines is not None and text.count("\n", 0, m.start()) + 1 not in only_lines:
            continue
        line_start = text.rfind("\n", 0, m.start()) + 1
        if "//" in text[line_start:m.start()]:
            continue  # commented out
        end = statement_end(text, m.start())
        out.append(text[pos:m.start()])
        out.append("{ simt::converge(); " + text[m.start():end] + " simt::converge(); }")
        pos = end
    out.append(text[pos:])
    return "".join(out)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--converge")]
    text = open(args[0]).read()
    # (line numbers of --converge-lines are those of the reference source: convergence first, the launch rewrite keeps lines)
    for a in sys.argv[1:]:
        if a == "--converge":
            text = add_convergence(text)
        elif a.startswith("--converge-lines="):  # --converge-lines=<source file name>:<line>,<line>,... (other files are left alone)
            name, lines = a.split("=")[1].split(":")
            if os.path.basename(args[0]) == name:
                text = add_convergence(text, set(int(x) for x in lines.split(",") if x))
    open(args[1], "w").write(rewrite(text))
