#!/usr/bin/env python3
"""Strip experiment-only preprocessor branches from a source file: every #ifdef / #ifndef / #if !defined(X) /
#if defined(X) whose macro name starts with one of the given prefixes is resolved as if the macro were NOT defined.
Tunables of the form `#ifndef X / #define X default / #endif` are kept (they define X themselves).
usage: unifdef_exp.py file prefix [prefix ...]   (rewrites the file in place, prints what it removed)"""
import re
import sys


def main():
    path, prefixes = sys.argv[1], tuple(sys.argv[2:])
    lines = open(path).read().split("\n")
    out, stack, removed = [], [], 0          # stack entries: (is_ours, currently_keeping, parent_keeping)
    i = 0
    while i < len(lines):
        ln = lines[i]
        st = ln.strip()
        m = re.match(r"#\s*(ifdef|ifndef)\s+(\w+)", st) or None
        m2 = re.match(r"#\s*if\s+(!?)\s*defined\s*\(\s*(\w+)\s*\)\s*(//.*)?$", st)
        keeping = all(s[1] for s in stack)
        if m or m2:
            if m:
                neg, name = m.group(1) == "ifndef", m.group(2)
            else:
                neg, name = m2.group(1) == "!", m2.group(2)
            ours = name.startswith(prefixes)
            if ours and neg and i + 1 < len(lines) and re.match(r"#\s*define\s+" + name + r"\b", lines[i + 1].strip()):
                ours = False                                    # a tunable with its default: keep as is
            if ours:
                stack.append((True, neg, keeping))             # macro undefined: #ifdef -> drop, #ifndef -> keep
                removed += 1
            else:
                stack.append((False, True, keeping))
                if keeping:
                    out.append(ln)
        elif re.match(r"#\s*if\b", st):
            stack.append((False, True, keeping))
            if keeping:
                out.append(ln)
        elif re.match(r"#\s*else\b", st) and stack:
            ours, k, pk = stack[-1]
            if ours:
                stack[-1] = (True, not k, pk)
            elif keeping:
                out.append(ln)
        elif re.match(r"#\s*endif\b", st) and stack:
            ours, k, pk = stack.pop()
            if not ours and all(s[1] for s in stack):
                out.append(ln)
        else:
            if keeping:
                out.append(ln)
        i += 1
    assert not stack, "unbalanced conditionals in %s" % path
    open(path, "w").write("\n".join(out))
    print("%s: %d experiment conditionals resolved" % (path, removed))


main()
