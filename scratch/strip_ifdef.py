"""Removes the preprocessor branches that depend on an UNDEFINED macro (the record of those experiments is scratch/*.patch).
usage: strip_ifdef.py FILE MACRO   -- handles `#if defined(M) && (...)`, `#if (defined(M) && (...)) || OTHER`, `#ifdef M`."""
import re, sys
path, macro = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
out, stack = [], []          # stack entries: [kind, keep_now]; kind 'strip' = ours (false condition), 'strip_or' = `(ours) || rest`, 'other'
def emitting():
    return all(k[1] for k in stack)
for ln in lines:
    s = ln.strip()
    m_if = re.match(r"#\s*(if|ifdef|ifndef)\b(.*)", s)
    if m_if:
        cond = m_if.group(2)
        if m_if.group(1) == "ifdef" and cond.split()[0] == macro:
            stack.append(["strip", False]); continue
        if m_if.group(1) == "if" and re.match(r"\s*defined\(%s\)\s*&&" % macro, cond):
            stack.append(["strip", False]); continue
        m_or = m_if.group(1) == "if" and re.match(r"\s*\(defined\(%s\)\s*&&\s*\([^)]*\)\)\s*\|\|\s*(.*?)(\s*//.*)?$" % macro, cond)
        if m_or:
            if emitting(): out.append(re.sub(r"#\s*if.*", "#if " + m_or.group(1), ln))
            stack.append(["other", True]); continue
        if emitting(): out.append(ln)
        stack.append(["other", True]); continue
    if re.match(r"#\s*(else|elif)\b", s):
        if stack[-1][0] == "strip":
            stack[-1][1] = True; continue
        if emitting(): out.append(ln)
        continue
    if re.match(r"#\s*endif\b", s):
        k = stack.pop()
        if k[0] == "strip": continue
        if emitting(): out.append(ln)
        continue
    if emitting(): out.append(ln)
assert not stack
open(path, "w").write("\n".join(out))
