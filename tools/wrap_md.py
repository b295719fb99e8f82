#!/usr/bin/env python3
"""Wraps a Markdown file to a column limit: paragraphs and list items are re-flowed, tables whose rows would exceed the limit are turned
into nested lists (one item per row, one sub-item per cell, the header cell as its label), code fences and headings are left alone.
usage: wrap_md.py IN OUT [columns = 140]"""
import re
import sys
import textwrap


def wrap(text, width, first="", rest=""):
    return textwrap.fill(text, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def cells(row):
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|"):
        row = row[:-1]
    return [c.strip() for c in re.split(r"(?<!\\)\|", row)]


def convert_table(block, width):
    head = cells(block[0])
    out = []
    for row in block[2:] if len(block) > 1 and re.match(r"^\s*\|?\s*:?-{2,}", block[1]) else block[1:]:
        cs = cells(row)
        out.append(wrap("* **%s**" % cs[0], width, "", "  "))
        for h, c in zip(head[1:], cs[1:]):
            if c:
                out.append(wrap("- %s: %s" % (h, c) if h else "- %s" % c, width, "  ", "    "))
    return out


def main(src, dst, width=140):
    lines = open(src).read().split("\n")
    out, i, fence = [], 0, False
    while i < len(lines):
        l = lines[i]
        if l.lstrip().startswith("```"):
            fence = not fence
            out.append(l); i += 1; continue
        if fence or not l.strip() or l.startswith("#"):
            if l.startswith("#") and len(l) > width:
                out.append(wrap(l, width, "", "  "))
            else:
                out.append(l)
            i += 1; continue
        if l.lstrip().startswith("|"):
            j = i
            while j < len(lines) and lines[j].lstrip().startswith("|"):
                j += 1
            block = lines[i:j]
            if max(len(b) for b in block) > width:
                out += convert_table(block, width)
            else:
                out += block
            i = j; continue
        # a paragraph or list item: this line plus its continuation lines (until a blank line, a new item, a table, a heading, a fence)
        m = re.match(r"^(\s*)((?:[-*+]|\d+[.)])\s+)?", l)
        indent, bullet = m.group(1), m.group(2) or ""
        para = [l.strip()]
        j = i + 1
        while j < len(lines):
            n = lines[j]
            if not n.strip() or n.lstrip().startswith(("|", "```", "#")) or re.match(r"^\s*(?:[-*+]|\d+[.)])\s+", n):
                break
            para.append(n.strip()); j += 1
        text = " ".join(para)
        if bullet:
            text = text[len(bullet.strip()):].strip() if text.startswith(bullet.strip()) else text
            out.append(wrap(text, width, indent + bullet, indent + " " * len(bullet)))
        else:
            out.append(wrap(text, width, indent, indent))
        i = j
    open(dst, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 140)
