"""Resolves `#if MACRO` / `#if MACRO == n` / `#if !MACRO` blocks for a macro of known VALUE, and drops its #ifndef/#define default block.
usage: resolve_macro.py FILE MACRO VALUE"""
import re, sys
path, macro, value = sys.argv[1], sys.argv[2], int(sys.argv[3])
lines = open(path).read().split("\n")
out, stack = [], []
def emitting():
    return all(k[1] for k in stack)
i = 0
while i < len(lines):
    ln = lines[i]; s = ln.strip()
    # default block: #ifndef M / #define M v / #endif
    if re.match(r"#\s*ifndef\s+%s\b" % macro, s) and re.match(r"#\s*define\s+%s\b" % macro, lines[i + 1].strip()) and re.match(r"#\s*endif", lines[i + 2].strip()):
        i += 3; continue
    m = re.match(r"#\s*if\s+(.*?)(\s*//.*)?$", s)
    if m and not re.match(r"#\s*if(n?def)", s):
        c = m.group(1).strip()
        known = None
        if c == macro: known = value != 0
        elif c == "!" + macro: known = value == 0
        else:
            mm = re.match(r"%s\s*==\s*(\d+)$" % macro, c)
            if mm: known = value == int(mm.group(1))
        if known is not None:
            stack.append(["ours", known, known]); i += 1; continue      # [kind, emit_now, any_taken]
        if emitting(): out.append(ln)
        stack.append(["other", True, True]); i += 1; continue
    if re.match(r"#\s*if(n?def)\b", s):
        if emitting(): out.append(ln)
        stack.append(["other", True, True]); i += 1; continue
    if re.match(r"#\s*else\b", s):
        if stack[-1][0] == "ours":
            stack[-1][1] = not stack[-1][2]; i += 1; continue
        if emitting(): out.append(ln)
        i += 1; continue
    if re.match(r"#\s*elif\b", s):
        assert stack[-1][0] != "ours", "elif on a resolved macro: %s" % ln
        if emitting(): out.append(ln)
        i += 1; continue
    if re.match(r"#\s*endif\b", s):
        k = stack.pop()
        if k[0] != "ours" and emitting(): out.append(ln)
        i += 1; continue
    if emitting(): out.append(ln)
    i += 1
assert not stack
open(path, "w").write("\n".join(out))
