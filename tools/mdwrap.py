"""Re-wrap a markdown file to a line width (default 118): paragraphs and list items are re-flowed with their indentation
(hanging indent for bullets), tables / headings / fenced code are left alone.  usage: python tools/mdwrap.py IN OUT [WIDTH]"""
import re
import sys
import textwrap

src, dst = sys.argv[1], sys.argv[2]
width = int(sys.argv[3]) if len(sys.argv) > 3 else 118
lines = open(src).read().split("\n")
out, para, in_code = [], [], False


def flush():
    global para
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*)((?:[*\-+]|\d+\.)\s+)?", first)
    indent, bullet = m.group(1), m.group(2) or ""
    text = " ".join(s.strip() for s in para)
    text = text[len(bullet):] if bullet and text.startswith(bullet.strip()) else text
    if bullet:
        text = re.sub(r"^(?:[*\-+]|\d+\.)\s+", "", " ".join(s.strip() for s in para))
    wrapped = textwrap.wrap(text, width=width, initial_indent=indent + bullet, subsequent_indent=indent + " " * len(bullet),
                            break_long_words=False, break_on_hyphens=False)
    out.extend(wrapped or [""])
    para = []


for ln in lines:
    s = ln.strip()
    if s.startswith("```"):
        flush(); in_code = not in_code; out.append(ln); continue
    if in_code:
        out.append(ln); continue
    if not s:
        flush(); out.append(""); continue
    if s.startswith("|") or s.startswith("#") or s.startswith("{"):
        flush(); out.append(ln); continue
    is_item = re.match(r"^\s*(?:[*\-+]|\d+\.)\s+", ln) is not None
    if is_item:
        flush(); para = [ln]; continue
    if para:
        # a continuation line: same paragraph unless its indentation drops below the paragraph's text indentation
        para.append(ln)
    else:
        para = [ln]
flush()
open(dst, "w").write("\n".join(out))
