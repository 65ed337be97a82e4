#!/usr/bin/env python3
"""Re-wraps the prose of a markdown file at 118 columns (tables, headings, code fences and blank lines are left alone; a paragraph
or list item is joined and wrapped with its own indentation).  python tools/wrap_md.py DESIGN.md HISTORY.md"""
import re
import sys
import textwrap

ITEM = re.compile(r'^(\s*(?:[*-]|\d+\.)\s+)')


def wrap(text, width=118):
    out, fence, para = [], False, []

    def flush():
        if not para:
            return
        first = para[0]
        m = ITEM.match(first)
        ind = m.group(1) if m else re.match(r'^\s*', first).group(0)
        body = " ".join([first[len(ind):].strip()] + [p.strip() for p in para[1:]])
        w = textwrap.wrap(body, width=width - len(ind), break_long_words=False, break_on_hyphens=False)
        sub = " " * len(ind)
        out.extend([(ind if i == 0 else sub) + x for i, x in enumerate(w)])
        para.clear()

    for line in text.split("\n"):
        if line.startswith("```"):
            flush()
            fence = not fence
            out.append(line)
            continue
        if fence or line.startswith("|") or line.startswith("#") or not line.strip():
            flush()
            out.append(line)
            continue
        if ITEM.match(line):
            flush()
        para.append(line)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        s = wrap(open(p).read())
        open(p, "w").write(s)
        print(p, "longest prose line:", max(len(ln) for ln in s.split("\n") if not ln.startswith("|")))
