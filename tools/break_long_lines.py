#!/usr/bin/env python3
"""Breaks source lines longer than a limit at a comma (or, in a #define, at a statement end) that sits inside brackets and outside string
literals -- a line break both C++ and Python accept there.  Lines it cannot break safely are reported and left alone.
usage: break_long_lines.py LIMIT FILE..."""
import sys


def safe_points(line, seps):
    """indices right after a separator at bracket depth >= 1 (or any depth for '; ' in macros), outside string / char literals"""
    pts, depth, q, i = [], 0, None, 0
    while i < len(line):
        c = line[i]
        if q:
            if c == "\\":
                i += 2; continue
            if c == q:
                q = None
        elif c in "\"'":
            q = c
        elif c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        else:
            for s, need_depth in seps:
                if line.startswith(s, i) and (depth >= 1 or not need_depth):
                    pts.append(i + len(s))
        i += 1
    return pts


def break_line(line, limit):
    macro = line.lstrip().startswith("#define") or line.rstrip().endswith("\\")
    indent = len(line) - len(line.lstrip())
    cont = " " * (indent + 4)
    out, cur = [], line
    while len(cur) > limit:
        tail = " \\" if macro else ""
        body = cur[:-2].rstrip() if macro and cur.rstrip().endswith("\\") else cur
        pts = [p for p in safe_points(body, ((", ", True), ("; ", False)) if macro else ((", ", True),)) if indent + 8 < p <= limit - len(tail) - 2]
        if not pts:
            return None
        p = pts[-1]
        out.append(body[:p].rstrip() + tail)
        cur = cont + body[p:].lstrip() + (" \\" if macro and cur.rstrip().endswith("\\") else "")
    out.append(cur)
    return out


def main(limit, files):
    for f in files:
        lines = open(f, encoding="utf-8").read().split("\n")
        res, changed = [], 0
        for n, l in enumerate(lines, 1):
            if len(l) <= limit or l.lstrip().startswith(("//", "#!", "# ", "* ", "/*")):
                res.append(l); continue
            b = break_line(l, min(limit, 160))
            if b is None:
                print("%s:%d: not broken (%d characters)" % (f, n, len(l)))
                res.append(l)
            else:
                res += b; changed += 1
        if changed:
            open(f, "w", encoding="utf-8").write("\n".join(res))
            print("%s: %d lines broken" % (f, changed))


if __name__ == "__main__":
    main(int(sys.argv[1]), sys.argv[2:])
