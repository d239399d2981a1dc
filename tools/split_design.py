"""One-off of round 6: DESIGN.md keeps the design as it is (sections 1 - 7), the per-round pages (0, 0b, 0c, 8, 8b) move to profiles/HISTORY.md,
and prose lines are wrapped at 160 columns (tables, code fences and headings are left alone; list items keep their indentation).
    python tools/split_design.py            # rewrites DESIGN.md and profiles/HISTORY.md in place"""
import os
import re
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def wrap(text, width=150):
    """paragraphs are re-flowed: a paragraph is a prose or list-item line and the lines that continue it (not blank, not a table row, heading, fence or new list item)"""
    item = re.compile(r"^(\s*)((?:[-*]|\d+\.)\s+)")
    special = lambda l: (not l.strip()) or l.startswith("|") or l.startswith("#") or l.startswith("```") or l.lstrip().startswith("|")
    out, fence, block = [], False, None

    def flush():
        nonlocal block
        if block is None:
            return
        indent, marker, words = block
        body = " ".join(w.strip() for w in words)
        out.extend(textwrap.wrap(body, width=width, initial_indent=indent + marker, subsequent_indent=indent + " " * len(marker), break_long_words=False, break_on_hyphens=False) or [indent + marker])
        block = None
    for line in text.split("\n"):
        if line.startswith("```"):
            flush(); fence = not fence; out.append(line); continue
        if fence or special(line):
            flush(); out.append(line); continue
        m = item.match(line)
        if m:
            flush(); block = (m.group(1), m.group(2), [line[m.end():]]); continue
        if block is None:
            ind = re.match(r"^\s*", line).group(0)
            block = (ind, "", [line[len(ind):]])
        else:
            block[2].append(line)
    flush()
    return "\n".join(out)


def sections(text):
    """[(heading line, body)] split at '## ' headings; the part before the first heading has heading None"""
    parts, cur_h, cur = [], None, []
    for line in text.split("\n"):
        if line.startswith("## "):
            parts.append((cur_h, "\n".join(cur)))
            cur_h, cur = line, []
        else:
            cur.append(line)
    parts.append((cur_h, "\n".join(cur)))
    return parts


if __name__ == "__main__":
    src = open(os.path.join(ROOT, "DESIGN.md")).read()
    keep, hist = [], []
    for h, body in sections(src):
        if h is None:
            continue                                             # the header is rewritten by hand
        if re.match(r"^## (0|0b|0c|8|8b)\. ", h):
            hist.append((h, body))
        else:
            keep.append((h, body))
    open(os.path.join(ROOT, "DESIGN.sections.tmp"), "w").write(wrap("\n".join(h + "\n" + b for h, b in keep)))
    open(os.path.join(ROOT, "HISTORY.sections.tmp"), "w").write(wrap("\n".join(h + "\n" + b for h, b in hist)))
    print("kept", [h[:30] for h, _ in keep], "history", [h[:30] for h, _ in hist])
